"""The callers of the E-step path that make up `smc++ estimate`, restated around the GPU engine:

  SMCModel            smcpp/model.py:97-240 with spline.Piecewise (smcpp/spline/piecewise.py, spline.py)
  Analysis            smcpp/analysis/base.py:21-191 + smcpp/analysis/analysis.py:21-152 — data pipeline, the two-stage
                      start (a bootstrap EM iteration on the un-binned data with ONE hidden state, then thinning / binning
                      and hidden states from the empirical TMRCA distribution, Gaussian-mixture quantiles with the
                      balanced-state fallback), inference manager set-up, Q / E_step / loglik, `dump()` = model.final.json
  EMOptimizer         smcpp/optimize/optimizers.py:16-260 (E-step / M-step loop, coordinate groups, +-3 log-unit bounds,
                      L-BFGS-B on -Q with the engine's gradients) with the plugins that change results:
                      LoglikelihoodMonitor (plugins/loglikelihood_monitor.py), ParameterOptimizer for rho
                      (plugins/parameter_optimizer.py), AnalysisSaver (plugins/analysis_saver.py)

Host-side Python like the reference's; every likelihood evaluation runs on the GPU through `smcpp_amd._smcpp`.
Derivatives: the model hands the engine the seed matrix d a_k / d y_j of its pieces with respect to the optimised
coordinates (`derivative_seeds`), the engine returns dQ/dy (what the reference carries in ad numbers)."""
from __future__ import annotations

import json
import logging
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.optimize

from . import _smcpp, data as D
from .model import Observable
from .posterior import balance_hidden_states

logger = logging.getLogger(__name__)

# smcpp/defaults.py
PIECES = 100
MINIMUM, MAXIMUM = 1e-4, 1e4
MIN_POP, MAX_POP = 1e-3, 1e3


class SMCModel(Observable):
    """Log population size `y` at `knots` (coalescent units), piecewise constant between knots and flat outside
    (spline.Piecewise), evaluated on `PIECES` log-spaced pieces for the engine."""
    NPOP = 1

    def __init__(self, knots, N0, pid=None):
        super().__init__()
        self._knots = np.array(knots, dtype=float)
        self.N0 = N0
        self.pid = pid
        self._y = np.zeros(len(self._knots))
        self._coords = []                        # coordinates the next Q() is differentiated with respect to

    # ---- smcpp/model.py:110-160 ----
    @property
    def knots(self):
        return self._knots

    @property
    def K(self):
        return len(self._knots)

    @property
    def s(self):
        k = self._knots
        return np.r_[k[0], np.diff(np.logspace(np.log10(k[0]), np.log10(k[-1]), PIECES))]

    def for_pop(self, pid):
        assert pid == self.pid
        return self

    def __len__(self):
        return len(self._y)

    def __getitem__(self, key):
        return self._y[key]

    def __setitem__(self, key, item):
        self._y[key] = item
        self.update_observers("model update")

    def randomize(self):
        self[:] = self._y + np.random.normal(0.0, 0.0001, size=len(self._y))

    def _piece_index(self, x):
        """spline.py:19-33 at order 0: index of the knot interval of each point (flat outside the knot range)."""
        ip = np.searchsorted(np.log(self._knots), np.log(np.atleast_1d(x)), side="right") - 1
        return np.clip(ip, 0, len(self._knots) - 1)

    def __call__(self, x):
        return np.exp(self._y[self._piece_index(x)])

    def stepwise_values(self):
        return np.clip(self(np.cumsum(self.s)), MIN_POP, MAX_POP)

    def regularizer(self):
        return float((np.diff(self._y, 2) ** 2).sum())

    def regularizer_gradient(self):
        g = np.zeros(len(self._y))
        d2 = np.diff(self._y, 2)
        for i, v in enumerate(d2):               # d2_i = y_i - 2 y_{i+1} + y_{i+2}
            g[i] += 2 * v; g[i + 1] -= 4 * v; g[i + 2] += 2 * v
        return g

    # ---- derivative plumbing (the role of dlist / ad numbers, smcpp/model.py:152-160, _smcpp.pyx:66-83) ----
    @property
    def dlist(self):
        return list(self._coords)

    def differentiate(self, coords):
        self._coords = list(coords)

    def derivative_seeds(self):
        """[pieces x len(coords)]: d a_k / d y_j = a_k where piece k reads coordinate j and is not clipped."""
        if not self._coords:
            return None
        ip = self._piece_index(np.cumsum(self.s))
        raw = np.exp(self._y[ip])
        live = (raw > MIN_POP) & (raw < MAX_POP)
        a = np.clip(raw, MIN_POP, MAX_POP)
        seeds = np.zeros((len(ip), len(self._coords)))
        for j, c in enumerate(self._coords):
            sel = (ip == c) & live
            seeds[sel, j] = a[sel]
        return seeds

    def to_dict(self):
        return {"class": "SMCModel", "knots": list(map(float, self._knots)), "N0": self.N0, "spline_class": "Piecewise",
                "y": list(map(float, self._y)), "pid": self.pid}

    @classmethod
    def from_dict(cls, d):
        assert d["class"] == "SMCModel" and d["spline_class"] == "Piecewise"
        r = cls(d["knots"], d["N0"], d["pid"])
        r._y[:] = d["y"]
        return r


@dataclass
class EstimateArgs:
    """The options of `smc++ estimate` that reach Analysis (smcpp/commands/estimate.py); defaults as the CLI's."""
    mu: float = 1.25e-8
    r: Optional[float] = None
    knots: int = 8
    w: int = 100
    thinning: Optional[int] = None
    em_iterations: int = 20
    algorithm: str = "L-BFGS-B"
    multi: bool = False
    regularization_penalty: float = 6.0
    lambda_: Optional[float] = None
    unfold: bool = False
    polarization_error: float = 0.5
    nonseg_cutoff: Optional[int] = None
    timepoints: Optional[tuple] = None
    xtol: float = 0.1
    ftol: float = 1e-4
    outdir: Optional[str] = None
    base: str = "model"
    cores: Optional[int] = None
    device: int = -1


class EMTerminationException(Exception):
    pass


class EMOptimizer:
    """E-step / M-step loop of smcpp/optimize/optimizers.py:163-197 with the result-changing plugins built in: the log-likelihood
    monitor (termination), ScaleOptimizer (common shift before every M-step), ParameterOptimizer("rho"), AnalysisSaver."""

    def __init__(self, analysis, algorithm, xtol, ftol, single, learn_rho, outdir=None, base="model"):
        self._analysis, self._algorithm, self._xtol, self._ftol, self._single = analysis, algorithm, xtol, ftol, single
        self._learn_rho = learn_rho
        self._outdir, self._base = outdir, base
        self._old_loglik = None
        self.logliks: List[float] = []

    def _coordinates(self):
        K = self._analysis.model.K
        return [[k] for k in range(K)][::-1] if self._single else [list(range(K))]

    def _f(self, x, coords):
        an = self._analysis
        an.model.differentiate(coords)
        an.model[coords] = x
        q, g = an.Q(gradient=True)
        if np.isinf(q):
            return np.inf, np.zeros(len(x))
        return -q, -g

    def _minimize(self, x0, coords, bounds):
        if len(coords) > 1:
            return scipy.optimize.minimize(self._f, x0, jac=True, args=(coords,), bounds=bounds, method=self._algorithm)
        # (the reference hands {'xtol', 'ftol'} to scipy's bounded scalar solver, optimizers.py:132-138: names that solver does
        # not know, so it runs at its default xatol = 1e-5; reproduced, a coarser stop would end the search 0.1 log-units early)
        res = scipy.optimize.minimize_scalar(lambda x: self._f(np.array([x]), coords)[0], bounds=bounds[0], method="bounded")
        res.x = np.array([res.x])
        return res

    def _post_estep(self, i):
        an = self._analysis
        if self._outdir:
            an.dump(os.path.join(self._outdir, ".{}.iter{}".format(self._base, i)))
        ll = an.loglik()
        self.logliks.append(ll)
        if self._old_loglik is None:
            logger.info("Loglik: %f", ll)
        else:
            improvement = (self._old_loglik - ll) / self._old_loglik
            logger.info("New loglik: %f\t(old: %f [%f%%])", ll, self._old_loglik, 100.0 * improvement)
            if improvement < 0:
                logger.warning("Loglik decreased")
            elif improvement < self._ftol:
                logger.info("Log-likelihood improvement < tol=%g; terminating", self._ftol)
                raise EMTerminationException()
        self._old_loglik = ll

    def _scale_step(self):
        """`ScaleOptimizer` (smcpp/optimize/plugins/scale_optimizer.py: enabled, registered by SMCPPOptimizer for every run incl.
        the bootstrap): before each M-step one bounded search over a common shift of all the model's coordinates."""
        an = self._analysis
        an.model.differentiate([])
        x0 = np.array(an.model[:], dtype=float)

        def f(alpha):
            an.model[:] = x0 + alpha
            return -float(an.Q())

        res = scipy.optimize.minimize_scalar(f, method="bounded", bounds=(-1, 1))
        an.model[:] = x0 + res.x

    def _update_rho(self):
        an = self._analysis
        lo, hi = an._theta / 100, 100 * an._theta

        def f(x):
            an.rho = x
            return -an.Q()

        res = scipy.optimize.minimize_scalar(f, method="bounded", bounds=(lo, hi))
        logger.info("New rho: %g", res.x)
        an.rho = float(res.x)

    def run(self, niter):
        an = self._analysis
        try:
            for i in range(niter):
                an.E_step()
                self._post_estep(i)
                # "pre M-step" observers (the reference keeps them in a WeakSet: their order is not defined there either)
                self._scale_step()
                if self._learn_rho:
                    self._update_rho()
                for coords in self._coordinates():
                    x0 = np.array(an.model[coords], dtype=float)
                    bounds = np.transpose([np.maximum(x0 - 3.0, np.log(MINIMUM)), np.minimum(x0 + 3.0, np.log(MAXIMUM))])
                    res = self._minimize(x0, coords, bounds)
                    an.model.differentiate([])
                    an.model[coords] = res.x
        except EMTerminationException:
            pass
        if self._outdir:
            an.dump(os.path.join(self._outdir, "{}.final".format(self._base)))


class Analysis:
    """One-population analysis: a data set, a model and an inference manager (smcpp/analysis/analysis.py:18-90).
    `contigs`: list of `smcpp_amd.data.Contig` (or paths of .smc(.gz) files)."""

    def __init__(self, contigs, args: EstimateArgs):
        self._args = args
        if args.cores is not None:
            _smcpp.set_num_threads(args.cores)
        # ---- base.py:24-47 ----
        self._N0 = 0.5e-4 / args.mu
        self._theta = 2.0 * self._N0 * args.mu
        self._rho = 2 * self._N0 * args.r if args.r is not None else self._theta
        self._penalty = 0.0
        self._niter = args.em_iterations
        self._pol = 0.0 if args.unfold else args.polarization_error
        # ---- base.py:49-61: load, long-run recoding, compression, span breaking, small-contig filter ----
        cs = [D.load_smc(c) if isinstance(c, str) else c for c in contigs]
        pops = {x for c in cs for x in c.pid}
        if len(pops) != 1:
            raise RuntimeError("Please use 'smc++ split' to estimate two-population models")
        self.populations = tuple(pops)
        cs = [D.recode_nonseg(c, args.nonseg_cutoff) for c in cs]
        for c in cs:
            c.data = D.compress_repeated_obs(c.data)
        cs = [p for c in cs for p in D.break_long_spans(c, 100000)]
        cs = D.drop_small_contigs(cs, 100000)
        self._watterson = D.watterson_theta(cs)
        self._mc_w = int(2e-3 * self._N0 / self._rho)
        self._mutation_counts = self._count_mutations(cs, self._mc_w)
        self._contigs = cs
        # ---- analysis.py:27-55: knots from a constant model at Watterson's size, bootstrap with ONE hidden state ----
        NeN0 = self._watterson / (2.0 * args.mu * self._N0)
        m = SMCModel([1.0], self._N0, None)
        m[:] = np.log(NeN0)
        # the reference's balance_hidden_states(model, M) returns M break points (M - 1 intervals) in generations and
        # is divided by 2 N0 here; this repository's takes the number of INTERVALS and returns coalescent units
        hs = balance_hidden_states(m, 1 + args.knots)
        t1 = tK = None
        if args.timepoints is not None:
            t1, tK = [x / 2 / self._N0 for x in args.timepoints]
        self._init_knots(hs, t1, tK)
        self._model = SMCModel(self._knots, self._N0, self.populations[0])
        self.hidden_states = np.array([0.0, np.inf])
        self._init_inference_manager()
        self.alpha = 1
        self._model[:] = np.log(NeN0)
        self._model.randomize()
        self._optimizer = EMOptimizer(self, args.algorithm, args.xtol, args.ftol, single=False, learn_rho=False)
        self._init_regularization()
        self._optimizer.run(1)
        self.bootstrap_loglik = self._optimizer.logliks[0]
        # ---- analysis.py:57-90: thin, bin, recode, compress; hidden states; the main model and manager ----
        out = []
        for c in self._contigs:
            thinning = args.thinning if args.thinning is not None else int(500 * np.log(2 + c.n[0]))
            d = D.thin_data(c.data, thinning) if thinning > 1 else c.data
            c2 = D.Contig(D.bin_observations(d, args.w, c.a), c.pid, c.n, c.a, c.fn)
            c2 = D.recode_monomorphic(c2)
            c2.data = np.ascontiguousarray(D.compress_repeated_obs(c2.data), dtype=np.int32)
            out.append(D.validate(c2))
        self._contigs = D.drop_uninformative_contigs(out)
        try:
            q = self._empirical_tmrca(2 * args.knots)
            hs = np.r_[0.0, q, np.inf]
            if not np.all(np.diff(hs) > 0):
                raise RuntimeError("quantiles are not increasing")
            self.hidden_state_source = "empirical TMRCA quantiles"
        except Exception as e:  # noqa: BLE001  (the reference falls back on ANY failure, analysis.py:68-72)
            logger.warning("Mixture model failed for setting hidden states. Error was: %s", e)
            hs = balance_hidden_states(m, 2 * args.knots - 1)
            self.hidden_state_source = "balanced"
        self.hidden_states = hs
        self._init_knots(hs, t1, tK)
        old = self._model
        self._model = SMCModel(self._knots, self._N0, self.populations[0])
        self._model[:] = np.log(old(self._knots))
        self._init_inference_manager()
        self.alpha = args.w
        self._optimizer = EMOptimizer(self, args.algorithm, args.xtol, args.ftol, single=not args.multi,
                                      learn_rho=args.r is None, outdir=args.outdir, base=args.base)
        self._init_regularization()

    # ---- helpers ----
    @staticmethod
    def _count_mutations(contigs, w):
        """CountMutations (data_filter.py:205-232): per-window heterozygosity scaled to w, windows more than half observed."""
        mc = []
        for c in contigs:
            nmiss, muts = D.windowed_mutation_counts(c, w)
            mc += [m * w / nm for m, nm in zip(muts, nmiss) if nm > 0.5 * w]
        return np.array(mc, dtype=float)

    def _init_knots(self, hs, t1, tK):
        """analysis.py:105-118."""
        knots = np.asarray(hs)[1:-1:2]
        mult = np.mean(knots[1:] / knots[:-1])
        k0 = knots[0]
        t = t1 or k0
        a = []
        while t < k0:
            a = np.r_[a, t]
            t *= mult
        knots = np.r_[a, knots]
        if tK is not None and tK > knots[-1]:
            knots = np.r_[knots, tK]
        self._knots = knots

    def _init_regularization(self):
        a = self._args
        self._penalty = a.lambda_ if a.lambda_ else abs(self.Q()) * (10 ** -a.regularization_penalty)

    def _empirical_tmrca(self, k):
        """analysis.py:136-152: quantiles of a k-component Gaussian mixture fitted to the windowed mutation counts."""
        import scipy.stats.mstats
        import sklearn.mixture
        X = self._mutation_counts
        gmm = sklearn.mixture.GaussianMixture(n_components=k).fit(X[:, None])
        Y = gmm.sample(n_samples=100000)[0]
        p = np.logspace(np.log10(0.01), np.log10(0.99), k)
        return np.asarray(scipy.stats.mstats.mquantiles(Y[Y > 0], p) / (2 * self._theta * self._mc_w))

    def _init_inference_manager(self):
        """base.py:89-121 for one population."""
        n = max(int(c.n[0]) for c in self._contigs)
        obs = [np.ascontiguousarray(c.data, dtype=np.int32) for c in self._contigs]
        self._im = _smcpp.PyOnePopInferenceManager(n, obs, self.hidden_states, self.populations, self._pol,
                                                   device=self._args.device)
        self._im.model = self._model
        self._im.theta = self._theta
        self._im.rho = self._rho
        self._im.alpha = self._alpha = 1

    # ---- base.py:123-191 ----
    def run(self, niter=None):
        self._optimizer.run(niter or self._niter)

    def Q(self, gradient=False):
        if not gradient:
            return float(np.sum(self._im.Q(separate=True))) - self._penalty * self._model.regularizer()
        q, jac = self._im.Q_with_gradient()
        coords = self._model.dlist
        g = jac.sum(axis=0) - self._penalty * self._model.regularizer_gradient()[coords]
        return float(q.sum()) - self._penalty * self._model.regularizer(), g

    def E_step(self):
        self._im.E_step()

    def loglik(self, reg=True):
        ll = self._im.loglik()
        return ll - self._penalty * self._model.regularizer() if reg else ll

    @property
    def model(self):
        return self._model

    @property
    def alpha(self):
        return self._alpha

    @alpha.setter
    def alpha(self, a):
        self._alpha = a
        self._im.alpha = a

    @property
    def rho(self):
        return self._rho

    @rho.setter
    def rho(self, r):
        self._rho = r
        self._im.rho = r

    @property
    def contigs(self):
        return self._contigs

    def dump(self, filename):
        """`model.final.json` (base.py:186-191)."""
        write_final_json(filename, self._theta, self._rho, self._alpha, self._model, {self.populations[0]: self.hidden_states})


def write_final_json(filename, theta, rho, alpha, model, hidden_states):
    """`BaseAnalysis.dump` (smcpp/analysis/base.py:186-191): theta / rho / alpha, `model.to_dict()` and the hidden states per
    population, sorted keys, indent 4 - the file `smc++ plot` / `smc++ posterior` read back."""
    d = {"theta": theta, "rho": rho, "alpha": alpha, "model": model.to_dict(),
         "hidden_states": {k: [float(x) for x in v] for k, v in hidden_states.items()}}
    with open(filename + ".json", "wt") as f:
        json.dump(d, f, sort_keys=True, indent=4)


# ---- a minimal EM driver on raw piece sizes (no hidden-state selection, no plugins): what the monotonicity tests drive ----
def em(contigs, n, hidden_states, a0, s, theta, rho, alpha=1.0, polarization_error=0.5, iterations=5,
       penalty=0.0, bounds=(1e-2, 1e2), device=-1, callback=None):
    """Returns `(model, logliks)`: `logliks[i]` is the log-likelihood at the parameters entering EM iteration i."""
    from .model import PiecewiseModel
    model = PiecewiseModel(np.array(a0, dtype=float), np.array(s, dtype=float), 1e4, "pop1")
    model.differentiable = True
    im = _smcpp.PyOnePopInferenceManager(n, contigs, hidden_states, ("pop1",), polarization_error, device=device)
    im.model = model
    im.theta = theta
    im.rho = rho
    im.alpha = alpha
    K = len(model.a)
    logliks = []

    def neg_q(x):
        model.a[:] = np.exp(x)
        model.update_observers("model update")
        q, jac = im.Q_with_gradient()
        f = -q.sum()
        g = -(jac.sum(axis=0)) * np.exp(x)          # chain rule for a = exp(x)
        if penalty > 0:
            d = np.diff(x)
            f += penalty * np.sum(d * d)
            gp = np.zeros(K)
            gp[:-1] -= 2 * penalty * d
            gp[1:] += 2 * penalty * d
            g = g + gp
        return f, g

    for it in range(iterations):
        im.E_step()
        logliks.append(im.loglik())
        if callback:
            callback(it, logliks[-1], model.a.copy())
        x0 = np.log(model.a)
        res = scipy.optimize.minimize(neg_q, x0, jac=True, method="L-BFGS-B",
                                      bounds=[(np.log(bounds[0]), np.log(bounds[1]))] * K,
                                      options={"maxiter": 50})
        model.a[:] = np.exp(res.x)
        model.update_observers("model update")
    im.E_step()
    logliks.append(im.loglik())
    return model, np.array(logliks)
