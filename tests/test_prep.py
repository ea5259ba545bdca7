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


@pytest.mark.parametrize("name", ["G1_M16_n4", "G2_M51_n6_longspans", "G3_M32_n10_2Mbp", "G4_M64_n20_2Mbp",
                                  "G6_M1_n4", "G7_M32_n8_chr11"])
def test_prep_reproduces_reference_parameters(name):
    g = load_golden(name)
    pi, T, E = _prep(g)
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-13)
    np.testing.assert_allclose(T, g["T"], rtol=1e-11, atol=1e-17)     # long double vs 256-bit MPFR 3x3 chain
    np.testing.assert_allclose(E, g["E"], rtol=1e-12)


def test_prep_m256_n50():
    import os
    from conftest import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "params_M256_n50.npz")))
    pi, T, E = _prep(g)
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-13)
    np.testing.assert_allclose(T, g["T"], rtol=1e-10, atol=1e-17)
    np.testing.assert_allclose(E, g["E"], rtol=1e-11)


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
    np.testing.assert_allclose(E, Eref, rtol=1e-12)


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
