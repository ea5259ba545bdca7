"""Host-side mirror of the reference's ``smcpp._smcpp`` module surface for the E-step path.

Same class names, constructor signatures, properties and error behaviour as ``smcpp/_smcpp.pyx``
(``_PyInferenceManager`` 122-308, ``PyOnePopInferenceManager`` 310-332, ``PyTwoPopInferenceManager`` 334-368),
implemented over the C ABI of ``include/smcpp_engine.h``.  The compute runs in HIP kernels on the MI355X;
nothing here falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _engine as E

aca = np.ascontiguousarray


def set_num_threads(k):
    """``_smcpp.pyx:61-64``."""
    E.lib().smcpp_set_num_threads(int(k))


class _PyInferenceManager:
    def _my_init(self, observations, hidden_states, im_id=None):
        self._im = None
        self._im_id = im_id
        self.seed = 1
        if len(observations) == 0:
            raise RuntimeError("Observations list is empty")
        hidden_states = np.asarray(hidden_states, dtype=np.float64)
        if not np.all(np.sort(hidden_states) == hidden_states):
            raise RuntimeError("Hidden states must be in ascending order")
        self._observations = [aca(ob, dtype=np.int32) for ob in observations]
        self._Ls = np.array([ob.shape[0] for ob in self._observations], dtype=np.int32)
        self._hs = aca(hidden_states)
        self._num_hmms = len(observations)
        self._model = None
        self._theta = self._rho = self._alpha = None

    def _ptrs(self):
        arr = (C.POINTER(C.c_int) * self._num_hmms)()
        for i, ob in enumerate(self._observations):
            arr[i] = ob.ctypes.data_as(C.POINTER(C.c_int))
        return arr

    def __del__(self):
        im = getattr(self, "_im", None)
        if im and E is not None:         # (at interpreter shutdown the module globals may already be gone)
            try:
                E.lib().smcpp_destroy(im)
            except Exception:  # noqa: BLE001
                pass
            self._im = None

    # ---- properties mirrored from _smcpp.pyx:157-183 ----
    @property
    def observations(self):
        return self._observations

    @property
    def theta(self):
        return self._theta

    @theta.setter
    def theta(self, v):
        self._theta = v
        E.check(E.lib().smcpp_set_theta(self._im, float(v)))

    @property
    def rho(self):
        return self._rho

    @rho.setter
    def rho(self, v):
        self._rho = v
        E.check(E.lib().smcpp_set_rho(self._im, float(v)))

    @property
    def alpha(self):
        return self._alpha

    @alpha.setter
    def alpha(self, v):
        self._alpha = v
        E.check(E.lib().smcpp_set_alpha(self._im, float(v)))

    def E_step(self, forward_backward_only=False):
        """``_smcpp.pyx:185-191``."""
        if None in (self.theta, self.rho, self.alpha):
            raise RuntimeError("theta / rho / alpha must be set")
        E.check(E.lib().smcpp_estep(self._im, int(bool(forward_backward_only))))

    @property
    def model(self):
        return self._model

    @model.setter
    def model(self, m):
        self._model = m
        if hasattr(m, "register"):
            m.register(self)
        self.update("model update")

    @property
    def save_gamma(self):
        return bool(E.lib().smcpp_get_save_gamma(self._im))

    @save_gamma.setter
    def save_gamma(self, sg):
        E.check(E.lib().smcpp_set_save_gamma(self._im, int(bool(sg))))

    @property
    def hidden_states(self):
        hs = np.zeros(len(self._hs))
        E.check(E.lib().smcpp_get_hidden_states(self._im, E.dptr(hs)))
        return list(hs)

    @hidden_states.setter
    def hidden_states(self, hs):
        hs = aca(hs, dtype=np.float64)
        if len(hs) != len(self._hs):
            raise RuntimeError("hidden states must be same size")
        E.check(E.lib().smcpp_set_hidden_states(self._im, len(hs), E.dptr(hs)))

    @property
    def M(self):
        return len(self._hs) - 1

    @property
    def keys(self):
        K = E.lib().smcpp_num_keys(self._im)
        kl = E.lib().smcpp_key_len(self._im)
        k = np.zeros((K, kl), dtype=np.int32)
        E.check(E.lib().smcpp_get_keys(self._im, E.iptr(k)))
        return k

    @property
    def emission_probs(self):
        keys = self.keys
        out = np.zeros((len(keys), self.M))
        E.check(E.lib().smcpp_get_emission_probs(self._im, E.dptr(out)))
        return {tuple(int(x) for x in k): out[i].copy() for i, k in enumerate(keys)}

    @property
    def gamma_sums(self):
        keys = self.keys
        ret = []
        for c in range(self._num_hmms):
            vals = np.zeros((len(keys), self.M))
            present = np.zeros(len(keys), dtype=np.uint8)
            E.check(E.lib().smcpp_get_gamma_sums(self._im, c, E.dptr(vals),
                                                 present.ctypes.data_as(C.POINTER(C.c_ubyte))))
            ret.append({tuple(int(x) for x in keys[k]): vals[k].copy() for k in range(len(keys)) if present[k]})
        return ret

    @property
    def gammas(self):
        ret = []
        for c in range(self._num_hmms):
            # sized from what the LAST E-step stored (the save_gamma flag may have been toggled since)
            ncol = int(E.lib().smcpp_gamma_cols(self._im, c))
            g = np.zeros((self.M, ncol))
            E.check(E.lib().smcpp_get_gamma(self._im, c, E.dptr(g)))
            ret.append(g)
        return ret

    def gamma_argmax(self, c=0):
        out = np.zeros(int(self._Ls[c]) + 1, dtype=np.int32)
        E.check(E.lib().smcpp_get_gamma_argmax(self._im, c, E.iptr(out)))
        return out

    @property
    def xisums(self):
        ret = []
        for c in range(self._num_hmms):
            x = np.zeros((self.M, self.M))
            E.check(E.lib().smcpp_get_xisum(self._im, c, E.dptr(x)))
            ret.append(x)
        return ret

    @property
    def pi(self):
        out = np.zeros(self.M)
        E.check(E.lib().smcpp_get_pi(self._im, E.dptr(out)))
        return out

    @property
    def transition(self):
        out = np.zeros((self.M, self.M))
        E.check(E.lib().smcpp_get_transition(self._im, E.dptr(out)))
        return out

    def Q(self, separate=False):
        """``_smcpp.pyx:277-301`` (values; derivative seeds are the M-step path, not built yet)."""
        q = np.zeros(4)
        E.check(E.lib().smcpp_q(self._im, E.dptr(q), None))
        if separate:
            return list(q)
        return float(q.sum())

    def Q_with_gradient(self):
        """The four Q terms and their forward-mode Jacobian [4 x nder] with respect to the derivative seeds handed
        to `set_params` (what `Q()` returns inside an ad number in the reference, `_smcpp.pyx:277-301`)."""
        q = np.zeros(4)
        nder = E.lib().smcpp_num_derivatives(self._im)
        jac = np.zeros((4, max(nder, 1)))
        E.check(E.lib().smcpp_q(self._im, E.dptr(q), E.dptr(jac)))
        return q, jac[:, :nder]

    def loglik(self):
        """``_smcpp.pyx:303-308``: sum over contigs."""
        return float(sum(self.logliks()))

    def logliks(self):
        out = np.zeros(self._num_hmms)
        E.check(E.lib().smcpp_loglik(self._im, E.dptr(out)))
        return out

    # ---- engine extensions ----
    def set_raw(self, pi, T, keys, Etab):
        pi = aca(pi, dtype=np.float64); T = aca(T, dtype=np.float64)
        keys = aca(keys, dtype=np.int32); Etab = aca(Etab, dtype=np.float64)
        E.check(E.lib().smcpp_set_raw(self._im, E.dptr(pi), E.dptr(T), len(keys), E.iptr(keys), E.dptr(Etab)))

    def set_chunking(self, rows_per_chunk=0, eps_alpha=0.0, eps_beta=0.0):
        E.check(E.lib().smcpp_set_chunking(self._im, int(rows_per_chunk), float(eps_alpha), float(eps_beta)))

    def set_warm_start(self, on=True):
        """Extension: reuse the previous E-step's converged chunk-boundary vectors as start vectors (see the header)."""
        E.check(E.lib().smcpp_set_warm_start(self._im, int(bool(on))))

    def set_prep_mode(self, host=False):
        """Where the cold preparation runs: device kernels (default) or the host routines (see the header)."""
        E.check(E.lib().smcpp_set_prep_mode(self._im, int(bool(host))))

    def device_index(self):
        """The HIP device this manager lives on."""
        return int(E.lib().smcpp_device(self._im))

    @property
    def debug(self):
        """`InferenceManager::debug` (`_smcpp.pxd:53`): declared by the reference, read by nothing in its C++."""
        return bool(E.lib().smcpp_get_debug(self._im))

    @debug.setter
    def debug(self, on):
        E.check(E.lib().smcpp_set_debug(self._im, int(bool(on))))

    def chain_mode(self):
        """Chain kernel family in use: 0 generic, 1 LDS-resident, 2 cooperative, 3 streamed operands, 4 lock-step (MFMA),
        5 scans over the semiseparable structure of the transition matrix (the other families are its fallback)."""
        return int(E.lib().smcpp_chain_mode(self._im))

    def describe(self):
        """The engine's environment switches and the plan this manager resolved (chain family, chunks, history passes, whether the
        stored passes of the last E-step ran their scans in float): `smcpp_describe`."""
        return E.describe(self._im)

    def last_timing(self):
        t = np.zeros(9)
        E.check(E.lib().smcpp_last_timing(self._im, E.dptr(t)))
        return dict(zip(["host_prep_ms", "chains_wall_ms", "forward_ms", "backward_ms", "stats_ms", "finalize_ms",
                         "device_total_ms", "fwd_passes", "bwd_passes"], t))

    def last_host_timing(self):
        t = np.zeros(4)
        E.check(E.lib().smcpp_last_host_timing(self._im, E.dptr(t)))
        return dict(zip(["cold_prep_ms", "eigensystems_ms", "staging_ms", "host_total_ms"], t))

    def stream(self):
        return E.lib().smcpp_stream(self._im)

    def set_global_keys(self, gkeys):
        gkeys = aca(gkeys, dtype=np.int32)
        E.check(E.lib().smcpp_set_global_keys(self._im, len(gkeys), E.iptr(gkeys)))

    def pack_stats(self):
        n = C.c_long(0)
        E.check(E.lib().smcpp_pack_stats(self._im, None, C.byref(n), 0))
        buf = np.zeros(n.value)
        E.check(E.lib().smcpp_pack_stats(self._im, E.dptr(buf), C.byref(n), 0))
        return buf

    def stats_len(self):
        n = C.c_long(0)
        E.check(E.lib().smcpp_pack_stats(self._im, None, C.byref(n), 0))
        return int(n.value)

    def pack_stats_device(self, device_ptr, sync=True):
        """Write the packed statistics into a device buffer of `stats_len()` doubles (e.g. `tensor.data_ptr()` of the
        fp64 tensor that is all-reduced over RCCL); returns after the kernel has finished, or - `sync=False` - right after
        the enqueue on the engine's stream (`stream()`), for a consumer that is ordered on that stream."""
        n = C.c_long(0)
        E.check(E.lib().smcpp_pack_stats(self._im, C.cast(C.c_void_p(int(device_ptr)), C.POINTER(C.c_double)),
                                         C.byref(n), 1 if sync else 2))
        return int(n.value)

    def unpack_stats_device(self, device_ptr, n):
        E.check(E.lib().smcpp_unpack_stats(self._im, C.cast(C.c_void_p(int(device_ptr)), C.POINTER(C.c_double)),
                                           int(n), 1))

    def unpack_stats(self, buf):
        buf = aca(buf, dtype=np.float64)
        E.check(E.lib().smcpp_unpack_stats(self._im, E.dptr(buf), len(buf), 0))

    # ---- the exchange issued by the engine itself through RCCL's C API (include/smcpp_engine.h: smcpp_rccl_*) ----
    @staticmethod
    def rccl_unique_id(libpath=None):
        out = C.create_string_buffer(128)
        E.check(E.lib().smcpp_rccl_unique_id((libpath or "").encode(), out))
        return out.raw

    def rccl_init(self, unique_id, rank, world, libpath=None):
        E.check(E.lib().smcpp_rccl_init(self._im, (libpath or "").encode(), bytes(unique_id), int(rank), int(world)))

    def rccl_exchange(self):
        """pack -> ncclAllReduce -> the reduced log-likelihood sum, all on the engine's stream; returns that sum"""
        v = np.zeros(1)
        E.check(E.lib().smcpp_rccl_exchange(self._im, E.dptr(v)))
        return float(v[0])

    def rccl_unpack(self):
        E.check(E.lib().smcpp_rccl_unpack(self._im))

    def rccl_fetch(self):
        out = np.zeros(self.stats_len())
        E.check(E.lib().smcpp_rccl_fetch(self._im, E.dptr(out), len(out)))
        return out


class PyOnePopInferenceManager(_PyInferenceManager):
    """``PyOnePopInferenceManager(n, observations, hidden_states, im_id, polarization_error)`` (_smcpp.pyx:310-332)."""

    def __init__(self, n, observations, hidden_states, im_id, polarization_error, device=-1):
        self._my_init(observations, hidden_states, im_id)
        im = C.c_void_p()
        E.check(E.lib().smcpp_create_onepop(int(n), self._num_hmms, E.iptr(self._Ls), self._ptrs(), len(self._hs),
                                            E.dptr(self._hs), C.c_double(polarization_error), int(device),
                                            C.byref(im)))
        self._im = im
        self._n = int(n)
        # sensible defaults (_smcpp.pyx:318-320)
        self.alpha = 1
        self.theta = 1e-4
        self.rho = 1e-4

    @property
    def pid(self):
        assert len(self._im_id) == 1
        return self._im_id[0]

    def update(self, message, *args, **kwargs):
        m = self._model.for_pop(self.pid) if hasattr(self._model, "for_pop") else self._model
        a = aca(np.asarray(m.stepwise_values(), dtype=np.float64))
        s = aca(np.asarray(m.s, dtype=np.float64))
        assert np.all(a > 0) and len(a) > 0
        seeds = m.derivative_seeds() if hasattr(m, "derivative_seeds") else None
        if seeds is None:
            E.check(E.lib().smcpp_set_params(self._im, len(a), E.dptr(a), None, 0, E.dptr(s)))
        else:
            da = aca(np.asarray(seeds, dtype=np.float64))
            E.check(E.lib().smcpp_set_params(self._im, len(a), E.dptr(a), E.dptr(da), da.shape[1], E.dptr(s)))


class PyTwoPopInferenceManager(_PyInferenceManager):
    """``PyTwoPopInferenceManager(n1, n2, a1, a2, observations, hidden_states, im_id, polarization_error)``
    (_smcpp.pyx:334-368).  `im.model = TwoPopulationModel(...)` feeds the distinguished model, both populations and
    the split to the engine's JointCSFS preparation; `set_raw` remains available."""

    def __init__(self, n1, n2, a1, a2, observations, hidden_states, im_id, polarization_error, device=-1):
        assert a1 + a2 == 2
        assert a1 in [1, 2]
        assert a2 in [0, 1]
        self._a1 = a1
        self._my_init(observations, hidden_states, im_id)
        im = C.c_void_p()
        E.check(E.lib().smcpp_create_twopop(int(n1), int(n2), int(a1), int(a2), self._num_hmms, E.iptr(self._Ls),
                                            self._ptrs(), len(self._hs), E.dptr(self._hs),
                                            C.c_double(polarization_error), int(device), C.byref(im)))
        self._im = im
        self.alpha = 1
        self.theta = 1e-4
        self.rho = 1e-4

    def update(self, message, *args, **kwargs):
        m = self._model
        pids = self._im_id
        dist = None if self._a1 == 1 else pids[0]              # both lineages apart -> special distinguished model
        dm = m.for_pop(dist)
        ms = [m.for_pop(p) for p in pids]
        arrs, seeds = [], []
        for q in (dm, ms[0], ms[1]):
            arrs.append((aca(np.asarray(q.stepwise_values(), dtype=np.float64)), aca(np.asarray(q.s, dtype=np.float64))))
            sd = q.derivative_seeds() if hasattr(q, "derivative_seeds") else None
            seeds.append(None if sd is None else aca(np.asarray(sd, dtype=np.float64)))
        nder = max([0] + [x.shape[1] for x in seeds if x is not None])
        ptr = lambda x: None if x is None else E.dptr(x)       # noqa: E731
        (ad, sd_), (a1, s1), (a2, s2) = arrs
        E.check(E.lib().smcpp_set_params_twopop(self._im, len(ad), E.dptr(ad), E.dptr(sd_), ptr(seeds[0]), len(a1),
                                                E.dptr(a1), E.dptr(s1), ptr(seeds[1]), len(a2), E.dptr(a2),
                                                E.dptr(s2), ptr(seeds[2]), C.c_double(float(m.split)), int(nder)))


class PyRateFunction:
    """Mirror of `PyRateFunction` (smcpp/_smcpp.pyx:370-399): cumulative hazard of a piecewise-constant model.
    Values are plain floats; `R_jac` / `average_coal_times_jac` additionally return the Jacobians with respect to the
    model's `derivative_seeds()` (the `.d()` parts of the reference's ad numbers)."""

    def __init__(self, model, hs):
        self._model = model
        self._hs = np.asarray(list(hs), dtype=np.float64)
        self._a = np.asarray([float(x) for x in model.stepwise_values()], dtype=np.float64)
        self._s = np.asarray(model.s, dtype=np.float64)

    def R(self, t):
        assert np.isfinite(t)
        return float(E.host_rate_function(self._a, self._s, [t])[0])

    def average_coal_times(self):
        if len(self._hs) < 2:
            return []
        return list(E.host_rate_function(self._a, self._s, [0.0], self._hs)[1])

    def _seeds(self):
        da = getattr(self._model, "derivative_seeds", lambda: None)()
        return np.eye(len(self._a)) if da is None else da

    def R_jac(self, t):
        R, dR = E.host_rate_function_jac(self._a, self._seeds(), self._s, [t])[:2]
        return float(R[0]), dR[0]

    def average_coal_times_jac(self):
        _, _, ct, dct = E.host_rate_function_jac(self._a, self._seeds(), self._s, [0.0], self._hs)
        return ct, dct

    def random_coal_times(self, t1, t2, K):
        seeds = np.random.randint(0, np.iinfo(np.int64).max, size=K, dtype=np.int64).astype(np.uint64)
        t, R = E.host_random_coal_times(self._a, self._s, t1, t2, seeds)
        return [[float(x), float(y)] for x, y in zip(t, R)]


def raw_sfs(model, n, t1, t2, below_only=False, jac=False):
    """Mirror of `raw_sfs` (smcpp/_smcpp.pyx:401-412): conditioned SFS [3, n+1] of the single hidden state [t1, t2)."""
    a = np.asarray([float(x) for x in model.stepwise_values()], dtype=np.float64)
    s = np.asarray(model.s, dtype=np.float64)
    if not jac:
        return E.host_raw_sfs(n, a, s, t1, t2, below_only)
    da = getattr(model, "derivative_seeds", lambda: None)()
    return E.host_raw_sfs(n, a, s, t1, t2, below_only, da=np.eye(len(a)) if da is None else da)


def joint_csfs(n1, n2, a1, a2, model, hidden_states, K=10):
    """Mirror of `joint_csfs` (smcpp/_smcpp.pyx:416-437, "used for testing purposes only"): list over hidden states of
    the joint conditioned SFS [(a1+1), (n1+1), (a2+1), (n2+1)] of a `TwoPopulationModel`."""
    assert (a1 == 2 and a2 == 0) or (a1 == a2 == 1)
    p1, p2 = model.for_pop(model.pids[0]), model.for_pop(model.pids[1])
    J = E.host_joint_csfs(n1, n2, a1, a2, np.asarray(list(hidden_states), dtype=np.float64),
                          (p1.stepwise_values(), p1.s), (p2.stepwise_values(), p2.s), float(model.split), K)
    return [J[m] for m in range(J.shape[0])]
