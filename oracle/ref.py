"""TEST INFRASTRUCTURE ONLY — ctypes view of ``oracle/_ref/libsmcpp_ref.so`` (the real reference sources
compiled where they lie under /root/reference by ``oracle/Makefile``; see ``ref_harness.cpp``).

Never imported by the product (``smcpp_amd/``).  Used (a) to pin the C restatement ``hmm_oracle.c``, (b) to emit
the golden vectors under ``tests/golden/`` and (c) as the ``cpu_baseline`` of ``bench.py`` (kind "reference").
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libsmcpp_ref.so")
_lib = None


def available() -> bool:
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        # libgmpxx lives only under /opt/conda/lib; load it by absolute path (no rpath: conda's libstdc++ is older
        # than the system one this process already uses)
        for dep in ("/opt/conda/lib/libgmpxx.so.4",):
            if os.path.exists(dep):
                C.CDLL(dep, mode=C.RTLD_GLOBAL)
        _lib = C.CDLL(_LIB_PATH)
        _lib.ref_last_error.restype = C.c_char_p
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def estep(pi, T, keys, E, obs, save_gamma=False, want_alpha=False, eig_key=-1):
    """Reference ``HMM::Estep`` on one contig with raw parameters. Returns a dict."""
    L_ = lib()
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    E = np.ascontiguousarray(E, dtype=np.float64)
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    M = pi.shape[0]
    K, keylen = keys.shape
    L = obs.shape[0]
    assert obs.shape[1] == 1 + keylen and E.shape == (K, M) and T.shape == (M, M)
    loglik = np.zeros(1)
    xisum = np.zeros((M, M))
    gamma = np.zeros((M, L + 1 if save_gamma else 1))
    gs_n = np.zeros(1, dtype=np.int32)
    gs_keys = np.zeros((K, keylen), dtype=np.int32)
    gs_vals = np.zeros((K, M))
    alpha = np.zeros((L + 1, M), dtype=np.float32) if want_alpha else None
    log_c = np.zeros(L + 1) if want_alpha else None
    q = np.zeros(4)
    eP = np.zeros((M, M)) if eig_key >= 0 else None
    ePi = np.zeros((M, M)) if eig_key >= 0 else None
    ed = np.zeros(M) if eig_key >= 0 else None
    esc = np.zeros(1) if eig_key >= 0 else None
    eim = np.zeros(1) if eig_key >= 0 else None
    rc = L_.ref_estep(M, K, keylen, _p(keys, C.c_int), _p(E, C.c_double), _p(pi, C.c_double), _p(T, C.c_double),
                      L, _p(obs, C.c_int), int(save_gamma), _p(loglik, C.c_double), _p(xisum, C.c_double),
                      _p(gamma, C.c_double), _p(gs_n, C.c_int), _p(gs_keys, C.c_int), _p(gs_vals, C.c_double),
                      _p(alpha, C.c_float), _p(log_c, C.c_double), _p(q, C.c_double),
                      int(eig_key), _p(eP, C.c_double), _p(ePi, C.c_double), _p(ed, C.c_double),
                      _p(esc, C.c_double), _p(eim, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    n = int(gs_n[0])
    out = dict(loglik=float(loglik[0]), xisum=xisum, gamma=gamma, q=q,
               gamma_sums={tuple(int(x) for x in gs_keys[i]): gs_vals[i].copy() for i in range(n)})
    if want_alpha:
        out["alpha_hat"] = alpha
        out["log_c"] = log_c
    if eig_key >= 0:
        out["eig"] = dict(P=eP, Pinv=ePi, d=ed, scale=float(esc[0]), max_imag=float(eim[0]))
    return out


def prep(a, s, hs, rho, theta, n=-1, raw=False):
    """Reference pi / transition / average coalescence times / conditioned SFS (after ``incorporate_theta``)."""
    L_ = lib()
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    M = len(hs) - 1
    pi = np.zeros(M)
    T = np.zeros((M, M))
    ct = np.zeros(M)
    csfs = np.zeros((M, 3, n + 1)) if n >= 0 else None
    rawc = np.zeros((M, 3, n + 1)) if (n >= 0 and raw) else None
    rc = L_.ref_prep(len(a), _p(a, C.c_double), _p(s, C.c_double), M, _p(hs, C.c_double),
                     C.c_double(rho), C.c_double(theta), int(n),
                     _p(pi, C.c_double), _p(T, C.c_double), _p(ct, C.c_double), _p(csfs, C.c_double),
                     _p(rawc, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    out = dict(pi=pi, T=T, avg_ct=ct, csfs=csfs)
    if raw:
        out["raw_csfs"] = rawc
    return out


def prep_jac(a, da, s, hs, rho, theta, n):
    """Reference values AND forward-mode Jacobians (w.r.t. the seeds ``da`` [Kp x nder] on the piece sizes) of pi,
    the transition matrix, the average coalescence times and the conditioned SFS after ``incorporate_theta``."""
    L_ = lib()
    a = np.ascontiguousarray(a, dtype=np.float64)
    da = np.ascontiguousarray(da, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    M = len(hs) - 1
    nder = da.shape[1]
    pi = np.zeros(M); dpi = np.zeros((M, nder))
    T = np.zeros((M, M)); dT = np.zeros((M, M, nder))
    ct = np.zeros(M); dct = np.zeros((M, nder))
    cs = np.zeros((M, 3, n + 1)); dcs = np.zeros((M, 3, n + 1, nder))
    rc = L_.ref_prep_jac(len(a), _p(a, C.c_double), _p(da, C.c_double), int(nder), _p(s, C.c_double), M,
                         _p(hs, C.c_double), C.c_double(rho), C.c_double(theta), int(n),
                         _p(pi, C.c_double), _p(dpi, C.c_double), _p(T, C.c_double), _p(dT, C.c_double),
                         _p(ct, C.c_double), _p(dct, C.c_double), _p(cs, C.c_double), _p(dcs, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    return dict(pi=pi, dpi=dpi, T=T, dT=dT, avg_ct=ct, davg_ct=dct, csfs=cs, dcsfs=dcs)


def rate(a, s, t, t1=0.0, t2=1.0, seeds=(), n=-1):
    """Reference ``PyRateFunction.R`` at ``t``, ``random_time(t1, t2, seed)`` (+ its R) per seed and, for ``n >= 0``,
    the ``compute_below`` part of the conditioned SFS of the state [t1, t2)."""
    L_ = lib()
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    t = np.ascontiguousarray(np.atleast_1d(t), dtype=np.float64)
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    R = np.zeros(len(t)); rt = np.zeros(len(seeds)); rR = np.zeros(len(seeds))
    below = np.zeros((3, n + 1)) if n >= 0 else None
    rc = L_.ref_rate(len(a), _p(a, C.c_double), _p(s, C.c_double), len(t), _p(t, C.c_double), _p(R, C.c_double),
                     C.c_double(t1), C.c_double(t2), len(seeds), _p(seeds, C.c_longlong), _p(rt, C.c_double),
                     _p(rR, C.c_double), int(n), _p(below, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    return dict(R=R, random_t=rt, random_R=rR, below=below)


def random_times_shared(a, s, t1, t2, K):
    """K draws of the reference's ``random_time(1., t1, t2, gen)`` from one default-seeded ``std::mt19937`` — the
    sequence ``JointCSFS`` uses (jcsfs.cpp:120-127).  Returns (t, R(t))."""
    L_ = lib()
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    t = np.zeros(K); R = np.zeros(K)
    rc = L_.ref_random_times_shared(len(a), _p(a, C.c_double), _p(s, C.c_double), C.c_double(t1), C.c_double(t2),
                                    int(K), _p(t, C.c_double), _p(R, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    return t, R


def q_jac(obs, keys, n, a, da, s, hs, rho, theta, alpha, polarization_error):
    """``HMM::Q`` values and gradients by the reference's own AD on one contig (see ``ref_q_jac`` in ref_harness.cpp).
    ``keys`` [K x 3] sorted as the manager holds them; ``da`` [Kp x nder] seeds.  Returns (q [4], jac [4 x nder], loglik)."""
    from . import prep_oracle
    L_ = lib()
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    a = np.ascontiguousarray(a, dtype=np.float64); s = np.ascontiguousarray(s, dtype=np.float64)
    da = np.ascontiguousarray(da, dtype=np.float64).reshape(len(a), -1)
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    M = len(hs) - 1
    nder = da.shape[1]
    tk = [tuple(int(x) for x in k) for k in keys]
    kind = np.zeros(len(tk), dtype=np.int32)
    off = [0]; idx = []; w = []
    need = [k for k in tk if not (k[2] == 0 and (k[0] == -1 or k[0] >= 0))]
    bins = prep_oracle.construct_bins(need, n, polarization_error) if need else {}
    for i, k in enumerate(tk):
        if k[2] == 0 and k[0] == -1:
            kind[i] = 1
        elif k[2] == 0 and k[0] >= 0:
            kind[i] = 2 + (k[0] % 2)
        else:
            for (aa, bb), p in bins[k].items():
                idx.append(aa * (n + 1) + bb); w.append(p)
        off.append(len(idx))
    off = np.array(off, dtype=np.int32)
    idx = np.array(idx if idx else [0], dtype=np.int32)
    w = np.array(w if w else [0.0], dtype=np.float64)
    q = np.zeros(4); jac = np.zeros((4, nder)); ll = np.zeros(1)
    rc = L_.ref_q_jac(M, len(tk), _p(keys, C.c_int), len(obs), _p(obs, C.c_int), int(n), len(a), _p(a, C.c_double),
                      _p(da, C.c_double), nder, _p(s, C.c_double), _p(hs, C.c_double), C.c_double(rho),
                      C.c_double(theta), C.c_double(alpha), _p(kind, C.c_int), _p(off, C.c_int), _p(idx, C.c_int),
                      _p(w, C.c_double), _p(q, C.c_double), _p(jac, C.c_double), _p(ll, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    return q, jac, float(ll[0])


def shift_or_truncate(a, s, t, truncate=False):
    """Reference ``shiftParams`` / ``truncateParams`` (src/common.cpp:63-98)."""
    L_ = lib()
    a = np.ascontiguousarray(a, dtype=np.float64); s = np.ascontiguousarray(s, dtype=np.float64)
    ao = np.zeros(len(a) + 2); so = np.zeros(len(a) + 2)
    K = C.c_int(0)
    rc = L_.ref_shift_or_truncate(int(bool(truncate)), len(a), _p(a, C.c_double), _p(s, C.c_double), C.c_double(t),
                                  C.byref(K), _p(ao, C.c_double), _p(so, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    return ao[:K.value].copy(), so[:K.value].copy()


def modified_moran(N, a, na):
    """Reference ``modified_moran_rate_matrix(N, a, na)`` (src/moran_eigensystem.cpp:31-52), dense."""
    L_ = lib()
    out = np.zeros((N + 1, N + 1))
    if L_.ref_modified_moran(int(N), int(a), int(na), _p(out, C.c_double)) != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    return out


def raw_csfs(a, s, hs, n, t=()):
    """Reference ``OnePopConditionedSFS(n).compute(eta)`` [M, 3, n+1] on hidden states ``hs`` and ``eta.R`` at ``t``."""
    L_ = lib()
    a = np.ascontiguousarray(a, dtype=np.float64); s = np.ascontiguousarray(s, dtype=np.float64)
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    M = len(hs) - 1
    out = np.zeros((M, 3, n + 1)) if n >= 0 else None
    R = np.zeros(len(t))
    rc = L_.ref_raw_csfs(len(a), _p(a, C.c_double), _p(s, C.c_double), M, _p(hs, C.c_double), int(n),
                         _p(out, C.c_double), len(t), _p(t, C.c_double), _p(R, C.c_double))
    if rc != 0:
        raise RuntimeError(L_.ref_last_error().decode())
    return out, R
