"""TEST INFRASTRUCTURE ONLY — Python front of the plain-C E-step restatement (``hmm_oracle.c``).

Never imported by the product (``smcpp_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it.  Pinned against the compiled reference (``oracle/ref.py``) and the committed golden
vectors (``tests/golden``).

Restated here (reference file:line):
  * key dictionary / ``ob_key``                     include/hmm.h:26, inference_manager.cpp:190-211
  * ``TransitionBundle::update`` eigensystems       src/transition_bundle.cpp:15-25, include/transition_bundle.h:9-30
    (LAPACK ``dgeev`` through numpy instead of Eigen's ``EigenSolver``; only real parts are kept, as there)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hmm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-o", _LIB_PATH, src, "-lm"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def key_ids(obs, keys=None):
    """Map rows to key ids.  ``keys`` (``[K x keylen]``) defaults to the sorted distinct keys of ``obs``
    (lexicographic = ``block_key::operator<``, block_key.h:47-56)."""
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    rowkeys = obs[:, 1:]
    if keys is None:
        keys = np.unique(rowkeys, axis=0)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    lut = {tuple(int(x) for x in k): i for i, k in enumerate(keys)}
    kid = np.fromiter((lut[tuple(int(x) for x in r)] for r in rowkeys), dtype=np.int32, count=len(rowkeys))
    return keys, kid


def eigensystem(T, b):
    """``eigensystem(EigenSolver(diag(b) Td^T))`` — P_r, Pinv_r, d_r, scale, max |imag d|."""
    A = b[:, None] * T.T
    d, P = np.linalg.eig(A)
    Pinv = np.linalg.inv(P)
    scale = float(np.abs(d).max())
    return (np.ascontiguousarray(P.real), np.ascontiguousarray(Pinv.real), np.ascontiguousarray(d.real), scale,
            float(np.abs(d.imag).max()))


def span_q(d_scaled, span):
    M = len(d_scaled)
    S = np.zeros((M, M))
    d_scaled = np.ascontiguousarray(d_scaled, dtype=np.float64)
    lib().oracle_span_q(M, int(span), _p(d_scaled, C.c_double), _p(S, C.c_double))
    return S


def estep(pi, T, keys, E, obs, save_gamma=False, eigs=None, want_beta=False):
    """Restated ``HMM::Estep`` + ``HMM::Q`` on one contig with raw parameters.

    ``keys`` / ``E`` list every key the manager knows (lexicographic order); rows of ``obs`` must use only those."""
    L_ = lib()
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    E = np.ascontiguousarray(E, dtype=np.float64)
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    keys, kid = key_ids(obs, keys)
    M = len(pi)
    K = len(keys)
    L = len(obs)
    span = np.ascontiguousarray(obs[:, 0])
    if np.any(span <= 0):
        raise RuntimeError("data are malformed: span <= 0")
    need = np.unique(kid[span > 1])
    eig_of_key = np.full(K, -1, dtype=np.int32)
    Ps, Pis, ds, scs = [], [], [], []
    max_imag = 0.0
    for j, k in enumerate(need):
        eig_of_key[k] = j
        es = eigs[int(k)] if eigs is not None else eigensystem(T, E[k])
        Ps.append(es[0]); Pis.append(es[1]); ds.append(es[2]); scs.append(es[3])
        max_imag = max(max_imag, es[4])
    nE = max(len(need), 1)
    P = np.ascontiguousarray(Ps, dtype=np.float64) if Ps else np.zeros((nE, M, M))
    Pinv = np.ascontiguousarray(Pis, dtype=np.float64) if Pis else np.zeros((nE, M, M))
    d = np.ascontiguousarray(ds, dtype=np.float64) if ds else np.zeros((nE, M))
    sc = np.ascontiguousarray(scs, dtype=np.float64) if scs else np.ones(nE)
    loglik = np.zeros(1)
    alpha = np.zeros((L + 1, M), dtype=np.float32)
    log_c = np.zeros(L + 1)
    xisum = np.zeros((M, M))
    gamma = np.zeros((L + 1, M)) if save_gamma else np.zeros((1, M))
    gsum = np.zeros((K, M))
    present = np.zeros(K, dtype=np.uint8)
    beta = np.zeros((L + 1, M)) if want_beta else None
    rc = L_.oracle_estep(M, K, _p(E, C.c_double), _p(pi, C.c_double), _p(T, C.c_double), L, _p(span, C.c_int),
                         _p(kid, C.c_int), _p(eig_of_key, C.c_int), _p(P, C.c_double), _p(Pinv, C.c_double),
                         _p(d, C.c_double), _p(sc, C.c_double), int(save_gamma), _p(loglik, C.c_double),
                         _p(alpha, C.c_float), _p(log_c, C.c_double), _p(xisum, C.c_double), _p(gamma, C.c_double),
                         _p(gsum, C.c_double), _p(present, C.c_ubyte), _p(beta, C.c_double))
    if rc != 0:
        raise RuntimeError("span")
    nb_pos = np.ascontiguousarray((keys[:, 2::3].sum(axis=1) > 0).astype(np.uint8))
    q = np.zeros(4)
    g0 = np.ascontiguousarray(gamma[0])
    L_.oracle_q(M, K, _p(E, C.c_double), _p(pi, C.c_double), _p(T, C.c_double), _p(g0, C.c_double),
                _p(gsum, C.c_double), _p(present, C.c_ubyte), _p(nb_pos, C.c_ubyte), _p(xisum, C.c_double),
                _p(q, C.c_double))
    return dict(loglik=float(loglik[0]), alpha_hat=alpha, log_c=log_c, xisum=xisum,
                gamma=(gamma.T.copy() if save_gamma else gamma[0][:, None].copy()),
                gamma_sums={tuple(int(x) for x in keys[k]): gsum[k].copy() for k in range(K) if present[k]},
                q=q, max_imag=max_imag, keys=keys, kid=kid, beta=beta)
