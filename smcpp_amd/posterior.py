"""The `smc++ posterior` product (reference `smcpp/commands/posterior.py:48-111` and
`smcpp/estimation_tools.py:170-197`): balanced hidden states, posterior decoding matrix and its argmax path."""
from __future__ import annotations

import numpy as np
import scipy.optimize

from . import _engine, _smcpp
from .model import PiecewiseModel


def balance_hidden_states(model, M):
    """Break points `[0, b_1, ..., b_{M-1}, inf)` (coalescent units) such that the probability of coalescing in each
    of the M intervals is equal under the model: `exp(-R(b_m)) = (M - m) / M`, root-found with Brent's method exactly
    as `estimation_tools.py:170-197` does (which is called with M+1 and returns generations = 2 N0 x these)."""
    a = np.asarray(model.stepwise_values(), dtype=float)
    s = np.asarray(model.s, dtype=float)
    ret = [0.0]
    for m in range(1, M):
        def f(t):
            return float(np.exp(-_engine.host_rate_function(a, s, t)[0]) - (M - m) / M)
        lo = hi = ret[-1]
        while f(lo) * f(hi) >= 0:
            hi = 2 * (hi + 1)
        ret.append(scipy.optimize.brentq(f, lo, hi))
    ret.append(np.inf)
    return np.array(ret)


def posterior(model, contigs, M, n, theta, rho, alpha=1.0, polarization_error=0.5, hidden_states=None, device=-1, a=None,
              start=None, end=None, thinning=1, return_manager=False):
    """Posterior decoding of each contig.  Returns `(hidden_states, gammas, sites, paths)`:
    `gammas[c]` is `[M, L+1]` with columns normalised to one (`posterior.py:102-106`), `sites[c]` the span column of the
    rows handed to the manager (missing row included) exactly as the reference stores it under `<file>_sites`
    (`posterior.py:109`: `obs[:, 0]`, one entry per row; cumulative positions are `np.cumsum` of it), `paths[c]` the
    argmax state per column computed on the device.  `return_manager=True` appends the inference manager (its device buffers
    stay allocated for as long as the caller keeps it) - nothing is kept otherwise.
    A missing row is prepended to every contig as the reference does (`posterior.py:83`); `start` / `end` keep the rows whose
    cumulative position lies in [start, end] (`posterior.py:76-82`: "only approximately picked out"), `thinning` > 1 thins every
    contig as `thin_dataset` does (`posterior.py:85-87`).
    TWO populations (`posterior.py:88-100`): `n = (n1, n2)` undistinguished and `a = (a1, a2)` distinguished lineages per
    population, rows of 7 columns, `model` a `TwoPopulationModel`; the hidden states are balanced with respect to the
    DISTINGUISHED lineages' model (`m.distinguished_model`, `posterior.py:60-62`)."""
    twopop = not np.isscalar(n) and len(n) == 2
    dist = model.model1 if twopop else model          # `distinguished_model` (smcpp/model.py:70-72,275-277)
    hs = balance_hidden_states(dist, M) if hidden_states is None else np.asarray(hidden_states, dtype=float)
    obs = []
    for c in contigs:
        d = np.asarray(c, dtype=np.int32)
        if start is not None or end is not None:
            pos = np.cumsum(d[:, 0])
            lb = 0 if start is None else start
            ub = pos[-1] if end is None else end
            d = d[(pos >= lb) & (pos <= ub)]
        miss = np.zeros((1, d.shape[1]), dtype=np.int32)
        miss[0, 0] = 1
        miss[0, 1::3] = -1
        obs.append(np.ascontiguousarray(np.vstack([miss, d])))
    if thinning > 1:
        from .data import thin_data
        obs = [np.ascontiguousarray(thin_data(o, thinning, 0)) for o in obs]
    if twopop:
        assert a is not None and len(a) == 2 and all(o.shape[1] == 7 for o in obs)
        pids = tuple(getattr(model, "pids", ("pop1", "pop2")))
        im = _smcpp.PyTwoPopInferenceManager(int(n[0]), int(n[1]), int(a[0]), int(a[1]), obs, hs, pids, polarization_error,
                                             device=device)
    else:
        nn = int(n if np.isscalar(n) else n[0])
        im = _smcpp.PyOnePopInferenceManager(nn, obs, hs, (getattr(model, "pid", "pop1"),), polarization_error, device=device)
    im.model = model
    im.theta = theta
    im.rho = rho
    im.alpha = alpha
    im.save_gamma = True
    im.E_step()
    gammas, sites, paths = [], [], []
    for c, g in enumerate(im.gammas):
        g = g / g.sum(axis=0, keepdims=True)
        gammas.append(g)
        sites.append(obs[c][:, 0].copy())
        paths.append(im.gamma_argmax(c))
    if return_manager:
        return hs, gammas, sites, paths, im
    return hs, gammas, sites, paths


def save_npz(path, hs, gammas, sites, names):
    """`.npz` layout of `smc++ posterior` (README.rst:348-372): `hidden_states`, `<file>`, `<file>_sites`."""
    out = {"hidden_states": hs}
    for nm, g, s in zip(names, gammas, sites):
        out[nm] = g
        out[nm + "_sites"] = s
    np.savez_compressed(path, **out)
