"""ctypes binding of the C ABI declared in ``include/smcpp_engine.h`` (``libsmcpp_engine.so``).

This is the binding a maintainer of the reference would write in ``_smcpp.pyx`` against the C ABI (see
INTEGRATION.md); here it is plain ctypes so the package needs no Cython build step."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build, _cabi

_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    """Load the engine library.  Fails loudly if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "The SMC++ MI355X engine has no CPU fallback.")
    L = C.CDLL(path)
    # every export gets its prototype: ints are coerced (numpy integer scalars included), `long` stays 64-bit, and a wrong
    # argument count is an error instead of stack garbage.  The prototypes are DERIVED from include/smcpp_engine.h (_cabi.py):
    # the header is the one table both bindings (this one and _smcpp_cy.pyx) are generated from.
    protos = _cabi.ctypes_prototypes()
    missing = [n for n in protos if not hasattr(L, n)]
    if missing:
        raise RuntimeError(f"libsmcpp_engine.so misses symbols include/smcpp_engine.h declares: {missing} - rebuild it")
    for name, (res, args) in protos.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


# every entry point include/smcpp_engine.h declares (the build check of __graft_entry__.build and the CPU tests walk this list)
EXPORTS = [name for name, _, _ in _cabi.declarations()]


def check(rc: int):
    if rc != 0:
        raise RuntimeError(lib().smcpp_last_error().decode())


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def host_set_csfs_direct(on):
    """Test hook: literal (reference-order) evaluation of the conditioned SFS instead of the factored one; returns the
    previous setting."""
    return bool(lib().smcpp_host_set_csfs_direct(int(bool(on))))


def host_chunk_counts(cost, rows, nslots, floor_cost=1):
    """Chunks per contig of the scan chains for `nslots` wavefront slots (host-only: `smcpp_host_chunk_counts`)."""
    cost = np.ascontiguousarray(cost, dtype=np.int64)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.zeros(len(cost), dtype=np.int32)
    check(lib().smcpp_host_chunk_counts(len(cost), cost.ctypes.data_as(C.POINTER(C.c_longlong)), iptr(rows),
                                        int(nslots), int(floor_cost), iptr(out)))
    return out


def host_eigensystem(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    P = np.zeros((n, n)); Pinv = np.zeros((n, n)); d = np.zeros(n)
    sc = C.c_double(0); mi = C.c_double(0)
    check(lib().smcpp_host_eigensystem(n, dptr(A), dptr(P), dptr(Pinv), dptr(d), C.byref(sc), C.byref(mi)))
    return P, Pinv, d, sc.value, mi.value


def host_eigensystem_team(A, threads):
    """`host_eigensystem` computed by `threads` cooperating threads (what the engine does for M >= 128)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    P = np.zeros((n, n)); Pinv = np.zeros((n, n)); d = np.zeros(n)
    sc = C.c_double(0); mi = C.c_double(0)
    check(lib().smcpp_host_eigensystem_team(n, dptr(A), int(threads), dptr(P), dptr(Pinv), dptr(d), C.byref(sc), C.byref(mi)))
    return P, Pinv, d, sc.value, mi.value


def host_prep_onepop(n, hs, polarization_error, a, s, theta, rho, alpha, keys):
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    M = len(hs) - 1
    K = len(keys)
    pi = np.zeros(M); T = np.zeros((M, M)); E = np.zeros((K, M))
    check(lib().smcpp_host_prep_onepop(int(n), len(hs), dptr(hs), float(polarization_error), len(a), dptr(a), dptr(s),
                                       float(theta), float(rho), float(alpha), K, iptr(keys), dptr(pi), dptr(T),
                                       dptr(E)))
    return pi, T, E


def host_prep_onepop_jac(n, hs, polarization_error, a, da, s, theta, rho, alpha, keys):
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    da = np.ascontiguousarray(da, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    M = len(hs) - 1
    K = len(keys)
    nder = da.shape[1]
    pi = np.zeros(M); T = np.zeros((M, M)); E = np.zeros((K, M))
    dpi = np.zeros((M, nder)); dT = np.zeros((M, M, nder)); dE = np.zeros((K, M, nder))
    check(lib().smcpp_host_prep_onepop_jac(int(n), len(hs), dptr(hs), float(polarization_error), len(a), dptr(a),
                                           dptr(da), int(nder), dptr(s), float(theta), float(rho), float(alpha), K,
                                           iptr(keys), dptr(pi), dptr(T), dptr(E), dptr(dpi), dptr(dT), dptr(dE)))
    return pi, T, E, dpi, dT, dE


def dev_prep_onepop(n, hs, polarization_error, a, s, theta, rho, alpha, keys, da=None, emulate=False):
    """Cold preparation with the conditioned SFS / emission table from the device kernels (``emulate``: the same kernel
    phases on the host).  Returns a dict: pi, T, E, sfs and, with ``da``, dpi, dT, dE, dsfs."""
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    M = len(hs) - 1
    K = len(keys)
    C_ = 3 * (int(n) + 1)
    nder = 0
    if da is not None:
        da = np.ascontiguousarray(da, dtype=np.float64)
        nder = da.shape[1]
    out = dict(pi=np.zeros(M), T=np.zeros((M, M)), E=np.zeros((K, M)), sfs=np.zeros((M, 3, int(n) + 1)))
    if nder:
        out.update(dpi=np.zeros((M, nder)), dT=np.zeros((M, M, nder)), dE=np.zeros((K, M, nder)),
                   dsfs=np.zeros((M, 3, int(n) + 1, nder)))
    check(lib().smcpp_dev_prep_onepop(int(n), len(hs), dptr(hs), float(polarization_error), len(a), dptr(a), dptr(da),
                                      int(nder), dptr(s), float(theta), float(rho), float(alpha), K, iptr(keys),
                                      int(bool(emulate)), dptr(out["pi"]), dptr(out["T"]), dptr(out["E"]),
                                      dptr(out.get("dpi")), dptr(out.get("dT")), dptr(out.get("dE")), dptr(out["sfs"]),
                                      dptr(out.get("dsfs"))))
    return out


def dev_q_emulate(n, hs, polarization_error, a, da, s, theta, rho, alpha, keys, g0, xi, gs):
    """Q's four terms and their gradient by the device kernel's phases run on the host (see the header)."""
    hs = np.ascontiguousarray(hs, dtype=np.float64); a = np.ascontiguousarray(a, dtype=np.float64)
    da = np.ascontiguousarray(da, dtype=np.float64); s = np.ascontiguousarray(s, dtype=np.float64)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    g0 = np.ascontiguousarray(g0, dtype=np.float64); xi = np.ascontiguousarray(xi, dtype=np.float64)
    gs = np.ascontiguousarray(gs, dtype=np.float64)
    nder = da.shape[1]
    val = np.zeros(4); jac = np.zeros((4, nder))
    check(lib().smcpp_dev_q_emulate(int(n), len(hs), dptr(hs), float(polarization_error), len(a), dptr(a), dptr(da), nder, dptr(s),
                                    float(theta), float(rho), float(alpha), len(keys), iptr(keys), dptr(g0), dptr(xi), dptr(gs),
                                    dptr(val), dptr(jac)))
    return val, jac


def host_rate_function(a, s, t, hs=None):
    """R(t) (and average coalescence times per hidden-state interval when ``hs`` is given)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    t = np.ascontiguousarray(np.atleast_1d(t), dtype=np.float64)
    R = np.zeros(len(t))
    if hs is None:
        check(lib().smcpp_host_rate_function(len(a), dptr(a), dptr(s), 0, None, len(t), dptr(t), dptr(R), None))
        return R
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    ct = np.zeros(len(hs) - 1)
    check(lib().smcpp_host_rate_function(len(a), dptr(a), dptr(s), len(hs), dptr(hs), len(t), dptr(t), dptr(R), dptr(ct)))
    return R, ct


def host_rate_function_jac(a, da, s, t, hs=None):
    """Values and Jacobians (w.r.t. the seeds ``da`` [K x nder]) of R(t) and, with ``hs``, of the average coalescence
    times: returns (R, dR) or (R, dR, ct, dct)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    da = np.ascontiguousarray(da, dtype=np.float64).reshape(len(a), -1)
    s = np.ascontiguousarray(s, dtype=np.float64)
    t = np.ascontiguousarray(np.atleast_1d(t), dtype=np.float64)
    nder = da.shape[1]
    R = np.zeros(len(t)); dR = np.zeros((len(t), nder))
    if hs is None or len(hs) < 2:
        check(lib().smcpp_host_rate_function_jac(len(a), dptr(a), dptr(da), nder, dptr(s), 0, None, len(t), dptr(t),
                                                 dptr(R), dptr(dR), None, None))
        return R, dR
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    ct = np.zeros(len(hs) - 1); dct = np.zeros((len(hs) - 1, nder))
    check(lib().smcpp_host_rate_function_jac(len(a), dptr(a), dptr(da), nder, dptr(s), len(hs), dptr(hs), len(t),
                                             dptr(t), dptr(R), dptr(dR), dptr(ct), dptr(dct)))
    return R, dR, ct, dct


def host_random_coal_times(a, s, t1, t2, seeds):
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    t = np.zeros(len(seeds)); R = np.zeros(len(seeds))
    check(lib().smcpp_host_random_coal_times(len(a), dptr(a), dptr(s), C.c_double(t1), C.c_double(t2), len(seeds),
                                             seeds.ctypes.data_as(C.POINTER(C.c_ulonglong)), dptr(t), dptr(R)))
    return t, R


def host_raw_sfs(n, a, s, t1, t2, below_only=False, da=None):
    """3 x (n+1) conditioned SFS of the single state [t1, t2) before ``incorporate_theta``; with ``da`` also its
    Jacobian [3, n+1, nder]."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    out = np.zeros((3, n + 1))
    if da is None:
        check(lib().smcpp_host_raw_sfs(int(n), len(a), dptr(a), None, 0, dptr(s), C.c_double(t1), C.c_double(t2),
                                       int(bool(below_only)), dptr(out), None))
        return out
    da = np.ascontiguousarray(da, dtype=np.float64).reshape(len(a), -1)
    nder = da.shape[1]
    dout = np.zeros((3, n + 1, nder))
    check(lib().smcpp_host_raw_sfs(int(n), len(a), dptr(a), dptr(da), nder, dptr(s), C.c_double(t1), C.c_double(t2),
                                   int(bool(below_only)), dptr(out), dptr(dout)))
    return out, dout


def host_joint_csfs(n1, n2, a1, a2, hs, model1, model2, split, K=10, da1=None, da2=None):
    """Joint CSFS per hidden state: array [M, a1+1, n1+1, a2+1, n2+1]; ``model*`` = (a, s).  With derivative seeds
    ``da1`` / ``da2`` ([K x nder], either may be None) also returns the Jacobian [..., nder]."""
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    a1v, s1v = (np.ascontiguousarray(x, dtype=np.float64) for x in model1)
    a2v, s2v = (np.ascontiguousarray(x, dtype=np.float64) for x in model2)
    out = np.zeros((len(hs) - 1, a1 + 1, n1 + 1, a2 + 1, n2 + 1))
    if da1 is None and da2 is None:
        check(lib().smcpp_host_joint_csfs(n1, n2, a1, a2, len(hs), dptr(hs), len(a1v), dptr(a1v), dptr(s1v), None,
                                          len(a2v), dptr(a2v), dptr(s2v), None, 0, C.c_double(split), int(K),
                                          dptr(out), None))
        return out
    nder = (da1 if da1 is not None else da2).shape[1]
    d1 = None if da1 is None else np.ascontiguousarray(da1, dtype=np.float64)
    d2 = None if da2 is None else np.ascontiguousarray(da2, dtype=np.float64)
    dout = np.zeros(out.shape + (nder,))
    check(lib().smcpp_host_joint_csfs(n1, n2, a1, a2, len(hs), dptr(hs), len(a1v), dptr(a1v), dptr(s1v),
                                      None if d1 is None else dptr(d1), len(a2v), dptr(a2v), dptr(s2v),
                                      None if d2 is None else dptr(d2), nder, C.c_double(split), int(K), dptr(out),
                                      dptr(dout)))
    return out, dout


def host_prep_twopop(n1, n2, a1, a2, hs, pol, dist, model1, model2, split, theta, rho, alpha, keys):
    hs = np.ascontiguousarray(hs, dtype=np.float64)
    ad, sd = (np.ascontiguousarray(x, dtype=np.float64) for x in dist)
    a1v, s1v = (np.ascontiguousarray(x, dtype=np.float64) for x in model1)
    a2v, s2v = (np.ascontiguousarray(x, dtype=np.float64) for x in model2)
    keys = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, 6)
    M = len(hs) - 1
    pi = np.zeros(M); T = np.zeros((M, M)); E = np.zeros((len(keys), M))
    check(lib().smcpp_host_prep_twopop(n1, n2, a1, a2, len(hs), dptr(hs), C.c_double(pol), len(ad), dptr(ad), dptr(sd),
                                       len(a1v), dptr(a1v), dptr(s1v), len(a2v), dptr(a2v), dptr(s2v), C.c_double(split),
                                       C.c_double(theta), C.c_double(rho), C.c_double(alpha), len(keys), iptr(keys),
                                       dptr(pi), dptr(T), dptr(E)))
    return pi, T, E


def set_option(name: str, value=None):
    """Set (or, `value=None`, remove) one SMCPP_* switch in the environment and make the engine re-read its option table
    (the engine parses the environment once per process: smcpp_amd/csrc/engine_options.hpp).  Affects managers built and
    E-steps run afterwards."""
    import os
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    lib().smcpp_reload_options()


def describe(im=None) -> dict:
    """The engine's switches and - for a manager handle - the plan it resolved (smcpp_describe)."""
    import json
    n = lib().smcpp_describe(im, None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().smcpp_describe(im, buf, n + 1)
    return json.loads(buf.value.decode())
