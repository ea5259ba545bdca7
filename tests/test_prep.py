"""Host-side cold preparation (smcpp_amd/csrc/prep.hpp) against the parameter files the
compiled reference emitted (tests/golden/params_*.npz, G*.npz) and, in the build container, the compiled reference live."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import prep_oracle, ref


def _prep(g, keys=None):
    from smcpp_amd import _engine
    keys = g["keys"] if keys is None else keys
    return _engine.host_prep_onepop(int(g["n"]), g["hs"], float(g["pol"]), g["a"], g["s"], float(g["theta"]),
                                    float(g["rho"]), float(g["alpha"]), keys)


GOLDENS = ["G1_M16_n4", "G2_M51_n6_longspans", "G3_M32_n10_2Mbp", "G4_M64_n20_2Mbp", "G6_M1_n4", "G7_M32_n8_chr11"]


@pytest.fixture
def csfs_direct():
    """Literal, term-by-term evaluation of the conditioned SFS (the reference's operation order) for one test."""
    from smcpp_amd import _engine
    prev = _engine.host_set_csfs_direct(True)
    yield
    _engine.host_set_csfs_direct(prev)


# The emission table is checked with  |E - E_ref| <= 3e-15 |E_ref| + 1e-18.  The absolute term is for the small
# entries (down to the 1e-10 floor of incorporate_theta, conditioned_sfs.cpp:100-148): they come out of the Moran
# back-transformation (|Uinv| up to 1e5 at n = 20, 1e14 at n = 50) by cancellation, so their low bits are rounding
# noise in the reference itself, and the factored evaluation rounds differently there (observed <= 4e-19 absolute,
# e.g. 1e-9 of a 2e-10 probability).  The literal evaluation below reproduces the reference's own rounding and keeps
# the purely relative 1e-12 of round 1.
@pytest.mark.parametrize("name", GOLDENS)
def test_prep_reproduces_reference_parameters(name):
    g = load_golden(name)
    pi, T, E = _prep(g)
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-13)
    np.testing.assert_allclose(T, g["T"], rtol=1e-11, atol=1e-17)     # long double vs 256-bit MPFR 3x3 chain
    np.testing.assert_allclose(E, g["E"], rtol=3e-15, atol=1e-18)


@pytest.mark.parametrize("name", GOLDENS)
def test_prep_literal_evaluation_reproduces_reference_parameters(name, csfs_direct):
    g = load_golden(name)
    pi, T, E = _prep(g)
    np.testing.assert_allclose(E, g["E"], rtol=1e-12)


def _m256():
    import os
    from conftest import GOLDEN
    return dict(np.load(os.path.join(GOLDEN, "params_M256_n50.npz")))


def test_prep_m256_n50():
    g = _m256()
    pi, T, E = _prep(g)
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-13)
    np.testing.assert_allclose(T, g["T"], rtol=1e-10, atol=1e-17)
    np.testing.assert_allclose(E, g["E"], rtol=5e-14, atol=1e-18)      # n = 50: |Uinv| reaches 1e14 (observed 8e-15)


def test_prep_m256_n50_literal(csfs_direct):
    g = _m256()
    pi, T, E = _prep(g)
    np.testing.assert_allclose(E, g["E"], rtol=1e-11)


@pytest.mark.parametrize("M,n,nder", [(1, 0, 0), (1, 3, 2), (6, 1, 0), (17, 7, 3), (40, 12, 0)])
def test_factored_csfs_equals_literal(M, n, nder):
    """The O(pieces n^2) prefix/suffix-sum evaluation against the O(pieces^2 n^2) literal one on random models,
    values and Jacobians, including a single state covering every piece (M = 1) and n = 0 / 1."""
    from smcpp_amd import _engine
    rng = np.random.default_rng(100 * M + n)
    Kp = 9
    a = np.exp(rng.normal(0, 1.0, Kp)); s = np.exp(rng.normal(-2.5, 1.0, Kp))
    hs = np.r_[0.0, np.sort(np.exp(rng.normal(-1.5, 1.5, M - 1))), np.inf] if M > 1 else np.array([0.0, np.inf])
    keys = [[0, 0, 0], [1, 0, 0], [-1, 0, 0]]
    if n > 0:
        keys += [[aa, b, n] for aa in (0, 1, 2) for b in range(n + 1) if not (aa == 0 and b == 0) and not (aa == 2 and b == n)]
    keys = np.array(sorted(map(tuple, keys)), dtype=np.int32)
    args = (n, hs, 0.2, a, s, 1e-2, 3e-3, 1.0, keys)
    da = rng.normal(0, 1, (Kp, nder)) if nder else None

    def run():
        if nder:
            return _engine.host_prep_onepop_jac(args[0], args[1], args[2], args[3], da, *args[4:])
        return _engine.host_prep_onepop(*args)

    fast = run()
    prev = _engine.host_set_csfs_direct(True)
    try:
        lit = run()
    finally:
        _engine.host_set_csfs_direct(prev)
    np.testing.assert_allclose(fast[2], lit[2], rtol=1e-12, atol=1e-18)
    if nder:
        np.testing.assert_allclose(fast[5], lit[5], rtol=1e-9, atol=1e-14 * max(1.0, np.abs(lit[5]).max()))


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
@pytest.mark.parametrize("M,n", [(8, 0), (8, 1), (8, 2), (24, 3), (51, 28)])
def test_prep_vs_reference_live_edge_sizes(M, n):
    from smcpp_amd import _engine
    hs = np.r_[0., np.logspace(-2, 1, M - 1), np.inf]
    a = np.array([1.0, 2.0, 0.5, 1.0, 3.0]); s = np.array([0.05, 0.2, 1.0, 1.0, 1.0])
    theta, rho, alpha, pol = 2.5e-2, 6e-3, 1.0, 0.3
    p = ref.prep(a, s, hs, rho, theta, n)
    keys = [[0, 0, 0], [1, 0, 0], [-1, 0, 0]]
    if n > 0:
        keys += [[0, b, n] for b in range(1, n + 1)] + [[1, b, n] for b in range(0, n + 1)]
        keys += [[2, b, n] for b in range(0, n)] + [[-1, min(1, n), n]]
        if n > 2:
            keys += [[0, 1, n - 1], [1, 0, n - 2], [2, n - 1, n - 1]]        # partially observed: hypergeometric lift
    keys = np.array(sorted(set(map(tuple, keys))), dtype=np.int32)
    ep = prep_oracle.emission_probs(keys, n, p["csfs"], p["avg_ct"], theta, alpha, pol)
    Eref = np.array([ep[tuple(int(x) for x in k)] for k in keys])
    pi, T, E = _engine.host_prep_onepop(n, hs, pol, a, s, theta, rho, alpha, keys)
    np.testing.assert_allclose(pi, p["pi"], rtol=1e-13)
    np.testing.assert_allclose(T, p["T"], rtol=1e-11, atol=1e-17)
    np.testing.assert_allclose(E, Eref, rtol=5e-14, atol=1e-18)


def test_known_answers_constant_size():
    """test/unit/test_bugs.py:18-33 intent: for a constant-size history R(t) = t, so pi_m = e^-t_m - e^-t_{m+1}, and
    every transition row sums to 1 - 1e-5/(M+1) (transition.cpp:247-252)."""
    from smcpp_amd import _engine
    hs = np.array([0.0, 0.5, 1.0, 2.0, np.inf])
    keys = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 2], [1, 1, 2]], dtype=np.int32)
    pi, T, E = _engine.host_prep_onepop(2, hs, 0.5, np.array([1.0]), np.array([1.0]), 1e-2, 1e-3, 1.0, keys)
    expect = np.array([1 - np.exp(-0.5), np.exp(-0.5) - np.exp(-1.0), np.exp(-1.0) - np.exp(-2.0), np.exp(-2.0)])
    np.testing.assert_allclose(pi, expect, rtol=1e-13)
    np.testing.assert_allclose(T.sum(axis=1), 1 - 1e-5 / 5, rtol=1e-12)
    assert np.all(E[2] == 1.0)                                           # missing emits 1
    np.testing.assert_allclose(E[0] + E[1], 1.0, rtol=1e-13)             # reduced keys: exp(-2 a theta E[T]) and complement


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
@pytest.mark.parametrize("M,n", [(8, 3), (32, 10)])
def test_jacobians_vs_reference_autodiff(M, n):
    """Forward-mode duals through rows A7-A10 against the reference's AutoDiffScalar (adouble) derivatives."""
    from smcpp_amd import _engine, synth
    hs = synth.hidden_states(M)
    a, s = synth.model_pieces(6)
    da = np.eye(6)[:, :4] + 0.25                       # arbitrary non-trivial seed matrix, nder = 4
    theta, rho, alpha, pol = 2.5e-2, 6.25e-3, 1.0, 0.5
    r = ref.prep_jac(a, da, s, hs, rho, theta, n)
    keys = np.array(sorted({(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, n), (1, 0, n), (1, n, n), (2, 1, n), (0, 1, n - 1)}),
                    dtype=np.int32)
    pi, T, E, dpi, dT, dE = _engine.host_prep_onepop_jac(n, hs, pol, a, da, s, theta, rho, alpha, keys)
    np.testing.assert_allclose(pi, r["pi"], rtol=1e-13)
    np.testing.assert_allclose(dpi, r["dpi"], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(T, r["T"], rtol=1e-11, atol=1e-17)
    np.testing.assert_allclose(dT, r["dT"], rtol=1e-8, atol=1e-15)
    # emission Jacobian: the assembly is linear in the CSFS and smooth in avg_ct; compare against the reference AD inputs
    ep0 = prep_oracle.emission_probs(keys, n, r["csfs"], r["avg_ct"], theta, alpha, pol)
    h = 1e-6
    for d in range(da.shape[1]):
        ep1 = prep_oracle.emission_probs(keys, n, r["csfs"] + h * r["dcsfs"][..., d], r["avg_ct"] + h * r["davg_ct"][:, d],
                                         theta, alpha, pol)
        dref = np.array([(ep1[tuple(k)] - ep0[tuple(k)]) / h for k in keys.tolist()])
        assert np.abs(dE[..., d] - dref).max() <= 1e-7 * max(np.abs(dref).max(), 1e-3)


# ---- module helpers next to the managers: PyRateFunction, raw_sfs, golden G8 ----
def _g8():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G8_prep_only.npz"))


@pytest.mark.parametrize("si", [0, 1, 2])
def test_rate_function_and_raw_sfs_golden(si):
    from smcpp_amd import _smcpp, _engine
    from smcpp_amd.model import PiecewiseModel
    g = _g8()
    a, s, hs, t = g[f"S{si}_a"], g[f"S{si}_s"], g[f"S{si}_hs"], g[f"S{si}_t"]
    m = PiecewiseModel(a, s, 1e4)
    eta = _smcpp.PyRateFunction(m, hs)
    np.testing.assert_allclose([eta.R(x) for x in t], g[f"S{si}_R"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(eta.average_coal_times(), g[f"S{si}_avg_ct"], rtol=1e-13)
    rt, rR = _engine.host_random_coal_times(a, s, 0.01, 0.5, g["seeds"].astype(np.uint64))
    np.testing.assert_allclose(rt, g[f"S{si}_random_t"], rtol=1e-13)
    np.testing.assert_allclose(rR, g[f"S{si}_random_R"], rtol=1e-13)
    assert np.all((rt > 0.01) & (rt < 0.5))
    for n in (0, 2, 10, 20):
        for iv, (t1, t2) in enumerate(g["intervals"]):
            want = g[f"S{si}_sfs_n{n}_i{iv}"]
            got = _smcpp.raw_sfs(m, n, t1, t2)
            np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-15 * np.abs(want).max())
            wb = g[f"S{si}_below_n{n}_i{iv}"]
            gb = _smcpp.raw_sfs(m, n, t1, t2, below_only=True)
            np.testing.assert_allclose(gb, wb, rtol=1e-11, atol=1e-15 * max(np.abs(wb).max(), 1e-300))


def test_reference_known_answers_rate_function_and_sfs():
    """The reference's own known-answer tests (test/unit/test_bugs.py:8-33), restated against this module."""
    import scipy.integrate
    from smcpp_amd import _smcpp
    from smcpp_amd.model import PiecewiseModel
    model1 = PiecewiseModel([1.0], [1.0], 1e4)
    eta = _smcpp.PyRateFunction(model1, [0.0, 1.0, 2.0, np.inf])
    assert eta.R(2.0) == 2.0
    n = 5
    raw = _smcpp.raw_sfs(model1, n - 2, 0.0, np.inf)
    undist = np.zeros(n)                                     # util.undistinguished_sfs: fold (a, b) -> a + b
    for i in range(3):
        for j in range(n - 1):
            if i + j < n:
                undist[i + j] += raw[i, j]
    assert np.allclose(undist[1:], 2.0 / np.arange(1, n))
    ts = [0.0, 0.5, 1.0, 2.0, np.inf]
    for t1, t2 in zip(ts[:-1], ts[1:]):
        ans = scipy.integrate.quad(lambda t: t * np.exp(-t), t1, t2)[0] / (np.exp(-t1) - np.exp(-t2))
        for nn in [0, 2, 10, 20]:
            np.testing.assert_allclose(_smcpp.raw_sfs(model1, nn, t1, t2).sum(axis=1)[1], 2.0 * ans)
    # random coalescence times stay inside the conditioning interval (test_bugs.py:8-16)
    s = np.diff(np.logspace(-2, 0.5, 33))
    m2 = PiecewiseModel(np.exp(np.linspace(-2.4, 3.6, 32) % 1.7 - 1.0), s, 1e4)
    for t, Rt in _smcpp.PyRateFunction(m2, []).random_coal_times(0.0, 0.02, 10):
        assert 0.0 < t < 0.02


def test_rate_function_jacobian_vs_finite_differences():
    """Mirror of test/unit/test_rate_function.py (AD of R(t) against one-sided differences), asserted here."""
    from smcpp_amd import _smcpp
    from smcpp_amd.model import PiecewiseModel
    K = 10
    s = np.diff(np.logspace(np.log10(.01), np.log10(3.), K + 1))
    a = np.exp(np.array([0.27, -1.05, 0.67, 0.31, -2.30, 0.26, 0.20, 0.50, 0.32, 0.50]))
    hs = np.concatenate([[0.], np.logspace(-2, 1, 10), [np.inf]])
    m = PiecewiseModel(a, s, 1e4)
    m.differentiable = True
    eta = _smcpp.PyRateFunction(m, hs)
    Rt, dR = eta.R_jac(1.08)
    ct, dct = eta.average_coal_times_jac()
    sf, dsf = _smcpp.raw_sfs(m, 6, 0.05, 0.9, jac=True)
    for k in range(K):
        ap = a.copy(); ap[k] += 1e-7
        mp = PiecewiseModel(ap, s, 1e4)
        ep = _smcpp.PyRateFunction(mp, hs)
        assert abs((ep.R(1.08) - Rt) * 1e7 - dR[k]) <= 1e-5 * max(1.0, abs(dR[k]))
        np.testing.assert_allclose((np.array(ep.average_coal_times()) - ct) * 1e7, dct[:, k], atol=2e-6)
        np.testing.assert_allclose((_smcpp.raw_sfs(mp, 6, 0.05, 0.9) - sf) * 1e7, dsf[:, :, k], atol=2e-6)


# ---- device cold preparation (smcpp_amd/csrc/prep_dev.hpp): the kernel phases run serially on the host ----
@pytest.mark.parametrize("fixture,nder", [("params_M32_n10.npz", 0), ("params_M64_n20.npz", 0), ("params_M64_n20.npz", 3),
                                          ("params_M256_n50.npz", 0), ("params_M256_n50.npz", 2)])
def test_device_preparation_phases_equal_host_preparation(fixture, nder):
    """The conditioned SFS / incorporate_theta / emission-table kernels of the device preparation are written as
    __host__ __device__ phases; run serially on the CPU they must reproduce the host preparation BIT FOR BIT (same
    operations in the same order, FMA contraction off), values and forward-mode Jacobians."""
    import os
    from smcpp_amd import _engine
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    n = int(g["n"])
    args = (n, g["hs"], float(g["pol"]), g["a"], g["s"], float(g["theta"]), float(g["rho"]), float(g["alpha"]), g["keys"])
    if nder == 0:
        pi, T, E = _engine.host_prep_onepop(*args)
        o = _engine.dev_prep_onepop(*args, emulate=True)
        assert np.array_equal(o["E"], E) and np.array_equal(o["pi"], pi) and np.array_equal(o["T"], T)
        if "csfs" in g.files:                      # the compiled reference's conditioned SFS after incorporate_theta
            np.testing.assert_allclose(o["sfs"], g["csfs"], rtol=3e-15, atol=1e-15)
        return
    da = np.random.default_rng(5).standard_normal((len(g["a"]), nder))
    pi, T, E, dpi, dT, dE = _engine.host_prep_onepop_jac(args[0], args[1], args[2], args[3], da, *args[4:])
    o = _engine.dev_prep_onepop(*args, da=da, emulate=True)
    assert np.array_equal(o["E"], E) and np.array_equal(o["dE"], dE)
    assert np.array_equal(o["dpi"], dpi)
    assert np.abs(dE).max() > 1e-4
    # the transition matrix of this entry: values from the double routines, derivative planes of its O(M) generators by the chain
    # rule over plain arrays (prep.hpp: transition_generators_jac) against the generic duals carried through every operation
    np.testing.assert_allclose(o["T"], T, rtol=1e-14, atol=0)
    sc = np.abs(dT).max()
    err = np.abs(o["dT"] - dT)
    assert err.max() <= 1e-13 * sc, err.max() / sc
    np.testing.assert_allclose(o["dT"], dT, rtol=1e-9, atol=1e-16 * sc)


@pytest.mark.parametrize("fixture,nder", [("params_M32_n10.npz", 3), ("params_M64_n20.npz", 5)])
def test_device_q_phases_equal_dense_evaluation(fixture, nder):
    """Q and its gradient as the device kernel forms them (transition matrix and its Jacobian from the O(M) generator planes,
    never materialised) against the plain dense evaluation sum w log x / sum (w / x) dx on the host preparation's pi, T, E and
    Jacobians."""
    import os
    from smcpp_amd import _engine
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    n = int(g["n"]); keys = g["keys"]; M = len(g["hs"]) - 1
    args = (n, g["hs"], float(g["pol"]), g["a"], g["s"], float(g["theta"]), float(g["rho"]), float(g["alpha"]), keys)
    rng = np.random.default_rng(9)
    da = rng.standard_normal((len(g["a"]), nder))
    pi, T, E, dpi, dT, dE = _engine.host_prep_onepop_jac(args[0], args[1], args[2], args[3], da, *args[4:])
    g0 = rng.random(M); xi = rng.random((M, M)) * np.exp(-np.abs(np.subtract.outer(np.arange(M), np.arange(M))))
    gs = rng.random((len(keys), M)) * 100.0
    gs[3] = 0.0                                                      # a key no contig holds
    val, jac = _engine.dev_q_emulate(args[0], args[1], args[2], args[3], da, *args[4:], g0, xi, gs)
    nb = keys[:, 2] > 0
    ref = np.array([np.sum(g0 * np.log(pi)), np.sum(gs[~nb] * np.log(E[~nb])), np.sum(gs[nb] * np.log(E[nb])), np.sum(xi * np.log(T))])
    rj = np.array([np.einsum("i,id->d", g0 / pi, dpi), np.einsum("km,kmd->d", gs[~nb] / E[~nb], dE[~nb]),
                   np.einsum("km,kmd->d", gs[nb] / E[nb], dE[nb]), np.einsum("ij,ijd->d", xi / T, dT)])
    np.testing.assert_allclose(val, ref, rtol=1e-12)
    for t in range(4):
        assert np.max(np.abs(jac[t] - rj[t])) <= 1e-11 * np.abs(rj[t]).max(), t
