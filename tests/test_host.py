"""CPU tests of the product's host side: the C ABI loads and exports every declared symbol, the host eigensolver,
the synthetic generator, and the algebra behind the batched sufficient statistics."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_c_abi_exports_every_declared_symbol():
    from smcpp_amd import _engine
    L = _engine.lib()
    hdr = open(os.path.join(ROOT, "include", "smcpp_engine.h")).read()
    declared = set(re.findall(r"\b(smcpp_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert set(_engine.EXPORTS) == declared            # (EXPORTS is derived from the header: _cabi.declarations)


def test_both_bindings_are_generated_from_the_header():
    """the ctypes table (`_engine.py`) and the `cdef extern` block of `_smcpp_cy.pyx` are both DERIVED from
    include/smcpp_engine.h (smcpp_amd/_cabi.py), so the two bindings cannot drift from the C ABI or from each other: the parsed
    declarations cover every `smcpp_*(` of the header, every one has a ctypes prototype with the right arity on the loaded library,
    and the extern block in the .pyx is byte for byte what the generator emits (`python -m smcpp_amd._cabi --write` refreshes it)."""
    from smcpp_amd import _cabi, _engine
    decls = _cabi.declarations()
    hdr = open(os.path.join(ROOT, "include", "smcpp_engine.h")).read()
    assert {d[0] for d in decls} == set(re.findall(r"\b(smcpp_[a-z0-9_]+)\s*\(", hdr))
    L = _engine.lib()
    for name, ret, args in decls:
        f = getattr(L, name)
        assert len(f.argtypes) == len(args), name
    assert _cabi.pyx_block_is_current(), "smcpp_amd/_smcpp_cy.pyx: extern block is stale - run python -m smcpp_amd._cabi --write"
    # spot checks of the type mapping
    p = _cabi.ctypes_prototypes()
    import ctypes as C
    assert p["smcpp_create_onepop"][1][3] == C.POINTER(C.POINTER(C.c_int)) and p["smcpp_stream"][0] == C.c_void_p
    assert p["smcpp_init_logger_cb"] == (None, [C.c_void_p]) and p["smcpp_last_error"] == (C.c_char_p, [])


def test_no_cpu_fallback_message_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smcpp_amd import _smcpp
    g = load_golden("G1_M16_n4")
    with pytest.raises(RuntimeError, match="no HIP device|HIP error"):
        _smcpp.PyOnePopInferenceManager(4, [g["obs"]], g["hs"], ("p",), 0.5)


@pytest.mark.parametrize("name", ["G1_M16_n4", "G4_M64_n20_2Mbp", "G2_M51_n6_longspans"])
def test_host_eigensystem_real(name):
    from smcpp_amd import _engine
    g = load_golden(name)
    M = len(g["pi"])
    k0 = int(np.argmax((g["keys"] == g["keys"][0] * 0).all(axis=1))) if (g["keys"] == 0).all(axis=1).any() else 0
    A = g["E"][k0][:, None] * g["T"].T
    P, Pinv, d, scale, imag = _engine.host_eigensystem(A)
    assert imag == 0.0
    assert np.abs(P @ np.diag(d) @ Pinv - A).max() < 1e-12
    assert np.abs(P @ Pinv - np.eye(M)).max() < 1e-12
    w = np.linalg.eigvals(A)
    assert np.abs(np.sort(d) - np.sort(w.real)).max() < 1e-12
    assert abs(scale - np.abs(w).max()) < 1e-13
    assert np.allclose(np.linalg.norm(P, axis=0), 1.0)         # unit-norm columns like EigenSolver


@pytest.mark.parametrize("n,threads", [(256, 8), (256, 3), (130, 5), (64, 2), (17, 3), (2, 2)])
def test_team_eigensystem_is_bit_identical_to_the_serial_one(n, threads):
    """nonsym_eig_team.hpp splits the EISPACK pipeline over a team of threads without changing the order of any
    floating-point operation on any matrix element: P, Pinv, d must be EQUAL to the serial routine's, for a real spectrum
    (the team path proper) and for a complex one (fallback to the serial routine on rank 0)."""
    from smcpp_amd import _engine
    rng = np.random.RandomState(n + threads)
    S = np.exp(-0.3 * np.abs(np.subtract.outer(np.arange(n), np.arange(n)))) * (0.5 + rng.rand(n, n)) + 1e-6
    S = np.triu(S) + np.triu(S, 1).T + 5 * np.eye(n)
    T = S / S.sum(axis=1, keepdims=True)                  # reversible chain: diag(e) T^T has a real spectrum
    A_real = (0.9 + 0.1 * rng.rand(n))[:, None] * T.T
    A_cplx = rng.rand(n, n) - 0.3
    for A, real in ((A_real, True), (A_cplx, False)):
        ref = _engine.host_eigensystem(A)
        got = _engine.host_eigensystem_team(A, threads)
        if real:
            assert ref[4] == 0.0
        for x, y in zip(ref, got):
            assert np.array_equal(np.asarray(x), np.asarray(y))


def test_host_eigensystem_complex_pairs():
    from smcpp_amd import _engine
    rng = np.random.RandomState(3)
    for n in (5, 12, 33):
        A = rng.rand(n, n) - 0.3
        P, Pinv, d, scale, imag = _engine.host_eigensystem(A)
        w, V = np.linalg.eig(A)
        assert np.abs(np.sort(d) - np.sort(w.real)).max() < 1e-10
        assert abs(imag - np.abs(w.imag).max()) < 1e-10
        assert abs(scale - np.abs(w).max()) < 1e-10
        # real parts of a unit-norm complex eigenbasis: P_r diag(d_r) Pinv_r differs from A, but for every real
        # eigenvalue the column must still be an eigenvector
        for j in range(n):
            if abs(d[j] - w.real[np.argmin(np.abs(w - d[j]))]) < 1e-10 and np.abs(w.imag[np.argmin(np.abs(w - d[j]))]) == 0:
                v = P[:, j]
                assert np.abs(A @ v - d[j] * v).max() < 1e-9


def test_synth_generator_is_deterministic():
    from smcpp_amd import synth
    a = synth.synth_contig(0, 2_000_000, 20)
    b = synth.synth_contig(0, 2_000_000, 20)
    assert np.array_equal(a, b)
    assert a.dtype == np.int32 and a.shape[1] == 4
    assert a[:, 0].sum() == 20_000                                   # bins
    assert synth.contig_crc(a) == 0xF59BCE86                         # frozen (commit the CRC)
    assert np.all(np.any(a[1:, 1:] != a[:-1, 1:], axis=1))           # run-length encoded
    g = np.load(os.path.join(ROOT, "tests", "golden", "params_M64_n20.npz"))
    full = synth.synth_contig(0, 100_000_000, 20)
    assert len(full) == int(g["rows_100mbp"]) and synth.contig_crc(full) == int(g["crc_100mbp"])
    assert synth.splitmix64_at(0, np.array([0]))[0] == np.uint64(0xE220A8397B1DCDAF)   # SplitMix64 reference value
    t = synth.synth_contig_twopop(2, 1_000_000, 10, 10)
    assert t.shape[1] == 7 and np.all(t[:, 4] == 0)


def test_batched_statistics_algebra():
    """The identities the GPU statistics kernels rely on: with alpha, beta fixed, the reference's
    per-row gamma / xi of a span>1 row equal omega * diag(P D (u w^T o S) Pinv) and omega * P (u w^T o S) Pinv B with
    omega = 1 / (scale * sum_j d~_j^span u_j w_j), so they can be summed per (span, key) group before the GEMMs."""
    from oracle import oracle
    g = load_golden("G1_M16_n4")
    pi, T, keys, E, obs = g["pi"], g["T"], g["keys"], g["E"], g["obs"]
    o = oracle.estep(pi, T, keys, E, obs, want_beta=True)
    al = o["alpha_hat"].astype(np.float64); be = o["beta"]; logc = o["log_c"]; kid = o["kid"]; span = obs[:, 0]
    M = len(pi); L = len(obs)
    X = np.zeros((M, M)); gs = np.zeros((len(keys), M))
    ell = np.arange(1, L + 1)
    m1 = span == 1
    a_prev = al[ell[m1] - 1]; a_cur = al[ell[m1]]; b = be[ell[m1]]; e = E[kid[m1]]
    p = (a_cur * b).sum(1); w1 = 1.0 / (np.exp(logc[ell[m1]]) * p)
    X += (a_prev * w1[:, None]).T @ (b * e)
    np.add.at(gs, kid[m1], a_cur * b / p[:, None])
    groups = {}
    for i in np.nonzero(~m1)[0]:
        groups.setdefault((int(span[i]), int(kid[i])), []).append(i)
    for (s, k), idx in groups.items():
        P, Pinv, d, sc, _ = oracle.eigensystem(T, E[k])
        idx = np.array(idx); dsc = d / sc
        U = al[idx] @ Pinv.T
        W = be[idx + 1] @ P
        om = 1.0 / (sc * ((dsc ** s)[None, :] * U * W).sum(1))
        Z = oracle.span_q(dsc, s) * ((U * om[:, None]).T @ W)
        Y = Z @ Pinv
        X += P @ Y * E[k][None, :]
        gs[k] += np.einsum("ij,j,ji->i", P, d, Y)
    X = np.maximum(X * T, 1e-20)
    assert np.max(np.abs(X - o["xisum"]) / np.abs(o["xisum"])) < 1e-10
    for k, v in o["gamma_sums"].items():
        ki = int(np.where((keys == k).all(1))[0][0])
        assert np.max(np.abs(gs[ki] - v)) <= 1e-10 * np.abs(v).max()


def test_scan_chain_chunks_never_exceed_the_wavefront_slots():
    """`smcpp_host_chunk_counts` (the allocation the engine cuts the scan chains' chunk lists with): every contig gets at least one
    chunk, the total never exceeds the slots (22 autosomes each rounded UP once gave 1 046 wavefronts for 1 024 SIMDs and the
    stragglers cost 27 % of the pass), no contig gets more chunks than rows, and the longest chunk is within one contig's
    rounding of the ideal."""
    from smcpp_amd import _engine as E
    from smcpp_amd import synth
    cost = np.array([int(x * 1e4) for x in synth.C3_LENGTHS_MBP], dtype=np.int64)        # binned: 10^4 positions per Mbp
    rows = (cost // 4).astype(np.int32)
    for nslots in (512, 1536, 3072):
        n = E.host_chunk_counts(cost, rows, nslots, 1024)
        assert n.min() >= 1 and n.sum() <= nslots and np.all(n <= rows)
        assert n.sum() == nslots                                        # long contigs: every slot is used
        longest = (cost / n).max()
        assert longest <= cost.sum() / nslots * (1.0 + 1.0 / n.min())
    # the floor: a small input gets few, long chunks
    small = np.array([5000, 3000], dtype=np.int64)
    n = E.host_chunk_counts(small, np.array([1200, 700], dtype=np.int32), 512, 1024)
    assert n.tolist() == [4, 2]                                         # 1250 / 1500 positions per chunk: none below the floor
    for tot in (1500, 9000, 40_000, 700_000):
        c2 = np.array([tot, tot // 3 + 1100], dtype=np.int64)
        n = E.host_chunk_counts(c2, np.array([tot // 4, tot // 12 + 300], dtype=np.int32), 512, 1024)
        assert np.all(c2 // n >= 1024) or np.all(n[c2 // n < 1024] == 1)
    # more contigs than slots: one chunk each
    many = np.full(40, 10_000, dtype=np.int64)
    n = E.host_chunk_counts(many, np.full(40, 2000, dtype=np.int32), 16, 1024)
    assert n.tolist() == [1] * 40
    # a contig cannot have more chunks than rows
    n = E.host_chunk_counts(np.array([10 ** 9, 10 ** 6], dtype=np.int64), np.array([3, 5000], dtype=np.int32), 512, 1024)
    assert n[0] <= 3 and n.sum() <= 512
