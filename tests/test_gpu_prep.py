"""Device cold preparation (smcpp_amd/csrc/prep_dev.hpp: conditioned SFS, incorporate_theta, emission table as HIP kernels)
against the host preparation (prep.hpp, itself pinned against the compiled reference by tests/test_prep.py) and against the
parameter files the compiled reference emitted.

Tolerance: the kernels execute the host routine's operations in the host routine's order with FMA contraction off, so the
only difference is the last bit of the device's exp / expm1 / log.  The Moran back-transformation amplifies that like any
other rounding (the reference's own literal and the factored host evaluation differ by 5e-16 ABSOLUTE): the
bars are 4e-15 absolute on the table and its conditioned SFS, 2e-9 relative on entries that sit on the 1e-10 floor.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

ABS_TOL = 4e-15
REL_TOL = 2e-9


def _args(g):
    return (int(g["n"]), g["hs"], float(g["pol"]), g["a"], g["s"], float(g["theta"]), float(g["rho"]), float(g["alpha"]),
            g["keys"])


@pytest.mark.parametrize("fixture", ["params_M32_n10.npz", "params_M64_n20.npz", "params_M256_n50.npz"])
def test_device_kernels_match_host_and_reference_parameters(fixture):
    from smcpp_amd import _engine
    g = np.load(os.path.join(GOLDEN, fixture))
    pi, T, E = _engine.host_prep_onepop(*_args(g))
    o = _engine.dev_prep_onepop(*_args(g))
    d = np.abs(o["E"] - E)
    print(fixture, "device vs host: max abs", d.max(), "max rel", (d / E).max())
    assert d.max() <= ABS_TOL and (d / E).max() <= REL_TOL
    # the compiled reference's table (tests/golden/make_golden.py) at the tolerance tests/test_prep.py holds the host to
    np.testing.assert_allclose(o["E"], g["E"], rtol=1e-8, atol=1e-15)
    if "csfs" in g.files:
        np.testing.assert_allclose(o["sfs"], g["csfs"], rtol=1e-8, atol=ABS_TOL)
    e = _engine.dev_prep_onepop(*_args(g), emulate=True)
    assert np.array_equal(e["E"], E)


@pytest.mark.parametrize("fixture,nder", [("params_M32_n10.npz", 4), ("params_M64_n20.npz", 16), ("params_M256_n50.npz", 3)])
def test_device_jacobians_match_host(fixture, nder):
    from smcpp_amd import _engine
    g = np.load(os.path.join(GOLDEN, fixture))
    a = _args(g)
    da = np.random.default_rng(11).standard_normal((len(g["a"]), nder))
    pi, T, E, dpi, dT, dE = _engine.host_prep_onepop_jac(a[0], a[1], a[2], a[3], da, *a[4:])
    o = _engine.dev_prep_onepop(*a, da=da)
    d = np.abs(o["E"] - E)
    assert d.max() <= ABS_TOL and (d / E).max() <= REL_TOL
    scale = np.abs(dE).max()
    dd = np.abs(o["dE"] - dE)
    print(fixture, "dE: max abs", dd.max(), "scale", scale)
    assert dd.max() <= 1e-12 * scale
    assert np.array_equal(o["dpi"], dpi)                                      # host routine either way
    assert np.max(np.abs(o["dT"] - dT)) <= 1e-13 * np.abs(dT).max()           # generator planes by the chain rule vs generic duals


@pytest.mark.parametrize("M,n", [(1, 4), (2, 1), (5, 2), (17, 7), (48, 28)])
def test_device_kernels_edge_sizes(M, n):
    """Ragged sizes: a single state [0, inf), one undistinguished lineage, hidden states that coincide with model break points."""
    from smcpp_amd import _engine, synth
    hs = synth.hidden_states(M) if M > 2 else np.array([0.0, np.inf] if M == 1 else [0.0, 0.3, np.inf])
    a, s = synth.model_pieces(6)
    if M > 2:
        hs = hs.copy()
        j = 1 + int(np.argmin(np.abs(hs[1:-1] - s[:2].sum())))
        hs[j] = s[:2].sum()                                    # on a break point of the model
    keys = np.array([[-1, 0, 0], [0, 0, 0], [1, 0, 0]] + [[aa, b, n] for aa in (0, 1) for b in range(n + 1) if not (aa == 0 and b == 0)]
                    + ([[0, 1, n - 1], [-1, 1, n]] if n >= 2 else []), dtype=np.int32)
    args = (n, hs, 0.3, a, s, 2e-2, 5e-3, 1.0, keys)
    pi, T, E = _engine.host_prep_onepop(*args)
    o = _engine.dev_prep_onepop(*args)
    d = np.abs(o["E"] - E)
    assert d.max() <= ABS_TOL and (d / E).max() <= REL_TOL


def _manager(g, obs):
    from smcpp_amd import _smcpp
    from smcpp_amd.model import PiecewiseModel
    im = _smcpp.PyOnePopInferenceManager(int(g["n"]), [obs], g["hs"], ("pop1",), float(g["pol"]))
    im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
    return im, PiecewiseModel(g["a"], g["s"], 1e4, pid="pop1")


@pytest.mark.parametrize("name", ["G3_M32_n10_2Mbp", "G4_M64_n20_2Mbp"])
def test_estep_with_device_preparation_matches_host_preparation_and_golden(name):
    """The model path (set_params -> E_step): device-prepared emission table read by the chains and the statistics straight
    from HBM, against the same E-step with the host preparation and against the compiled reference's outputs."""
    g = load_golden(name)
    obs = np.ascontiguousarray(g["obs"], dtype=np.int32)
    res = {}
    for host in (False, True):
        im, model = _manager(g, obs)
        im.set_prep_mode(host)
        im.model = model
        im.E_step()
        res[host] = (im.loglik(), im.xisums[0], np.array(im.Q(separate=True)), im.emission_probs)
        assert im.chain_mode() == 5
    ll_d, xs_d, q_d, ep_d = res[False]
    ll_h, xs_h, q_h, ep_h = res[True]
    assert abs(ll_d - ll_h) <= 1e-12 * abs(ll_h)
    assert np.max(np.abs(xs_d - xs_h)) <= 1e-9 * np.abs(xs_h).max()
    assert np.all(np.abs(q_d - q_h) <= 1e-10 * np.abs(q_h))
    for k, v in ep_h.items():
        assert np.max(np.abs(ep_d[k] - v)) <= ABS_TOL
    assert abs(ll_d - float(g["loglik"])) <= 1e-6 * abs(float(g["loglik"]))
    assert np.all(np.abs(q_d - g["q"]) <= 5e-6 * np.abs(g["q"]))


@pytest.mark.parametrize("fixture,rows", [("params_M32_n10.npz", 20000), ("params_M64_n20.npz", 20000), ("params_M256_n50.npz", 4000)])
def test_scan_chains_take_their_operator_from_the_generators(engine_opt, fixture, rows):
    """Round 6: on the device-prepared model path the M x M transition matrix is expanded only when somebody reads it - the scan chains
    take their O(M) operator straight from the generators the preparation computed (`ss_generators_from_tgen`), the expansion and its
    upload happen behind the chains' launches.  SMCPP_T_LAZY=0 expands first and derives / CHECKS the generators entry by entry from
    the expanded matrix (rounds 3-5).  Same E-step either way: the log-likelihood to 1e-13, statistics and Q to 1e-11, and the matrix
    the getter hands out afterwards is bit for bit the same."""
    from smcpp_amd import synth
    g = np.load(os.path.join(GOLDEN, fixture))
    obs = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, int(g["n"]))[:rows], dtype=np.int32)
    res = {}
    for lazy in ("0", "1"):
        engine_opt("SMCPP_T_LAZY", lazy)
        im, model = _manager(g, obs)
        for _ in range(2):                      # (the second E-step runs with what the first one left behind)
            im.model = model
            im.E_step()
        res[lazy] = (im.loglik(), im.xisums[0], np.array(im.Q(separate=True)), im.gamma_sums[0], im.transition)
        assert im.chain_mode() == 5
    (ll0, x0, q0, g0, T0), (ll1, x1, q1, g1, T1) = res["0"], res["1"]
    print(fixture, "lazy vs expanded-first: loglik", abs(ll1 - ll0) / abs(ll0), "xi sums", np.max(np.abs(x1 - x0) / np.abs(x0)))
    assert abs(ll1 - ll0) <= 1e-13 * abs(ll0)
    assert np.max(np.abs(x1 - x0) / np.abs(x0)) <= 1e-11
    assert np.all(np.abs(q1 - q0) <= 1e-11 * np.abs(q0))
    for k, v in g0.items():
        np.testing.assert_allclose(g1[k], v, rtol=1e-11, atol=1e-13 * np.abs(v).max())
    assert np.array_equal(T0, T1)


def test_q_gradient_with_device_preparation():
    """Q(val, jac) after a device preparation with derivative seeds equals the host-prepared one (G10 pins the latter against
    the reference's own automatic differentiation)."""
    from smcpp_amd import _engine
    g = load_golden("G4_M64_n20_2Mbp")
    obs = np.ascontiguousarray(g["obs"], dtype=np.int32)
    out = {}
    for host in (False, True):
        im, model = _manager(g, obs)
        im.set_prep_mode(host)
        im.model = model
        im.E_step()
        K = len(g["a"])
        da = np.eye(K)
        _engine.check(_engine.lib().smcpp_set_params(im._im, K, _engine.dptr(np.ascontiguousarray(g["a"])), _engine.dptr(da), K,
                                                     _engine.dptr(np.ascontiguousarray(g["s"]))))
        val = np.zeros(4); jac = np.zeros((4, K))
        _engine.check(_engine.lib().smcpp_q(im._im, _engine.dptr(val), _engine.dptr(jac)))
        out[host] = (val, jac)
    assert np.all(np.abs(out[False][0] - out[True][0]) <= 1e-10 * np.abs(out[True][0]))
    sc = np.abs(out[True][1]).max()
    assert np.max(np.abs(out[False][1] - out[True][1])) <= 1e-9 * sc
    assert sc > 1.0
