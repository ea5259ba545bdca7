"""More than 256 hidden states (round 5: 256 < M <= 512 on the scan chains with eight states per lane and the eigen-free statistics;
the reference has no limit, src/inference_manager.cpp:21-54).

  * M = 300 on 1 200 rows and M = 512 on 160 rows of the synthetic contig against the C restatement of hmm.cpp (oracle/) fed with the
    engine's own prepared parameters, with chunks short enough that the chunk-parallel fixed point iterates (the restatement follows
    the reference's 2 M^3 flops per span > 1 row on one core: 2 000 rows at M = 512 would take minutes of the suite's time);
  * M = 512 on 1 000 rows against golden G21 = the COMPILED reference (tests/golden/make_golden_m512.py), through `im.model = ...`:
    the engine's own cold preparation on the device;
  * what is not built beyond 256 fails loudly: save_gamma, a transition matrix without the reference's structure.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

LL_TOL = 1e-6
STAT_TOL = 5e-6


def _manager(M, n, obs, chunk=0):
    from smcpp_amd import _smcpp, synth
    from smcpp_amd.model import PiecewiseModel
    a, s = synth.model_pieces()
    im = _smcpp.PyOnePopInferenceManager(n, [obs], synth.hidden_states(M), ("pop1",), 0.5)
    im.model = PiecewiseModel(a, s, 1e4, "pop1")
    im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
    if chunk:
        im.set_chunking(chunk)
    return im


@pytest.mark.parametrize("M,rows,chunk", [(300, 1200, 300), (512, 160, 60)])
def test_more_than_256_states_vs_oracle(M, rows, chunk):
    from oracle import oracle
    from smcpp_amd import synth
    n = 10
    obs = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, n)[:rows], dtype=np.int32)
    im = _manager(M, n, obs, chunk=chunk)
    im.E_step()
    assert im.chain_mode() == 5 and im.M == M
    keys = im.keys
    ep = im.emission_probs
    Etab = np.array([ep[tuple(k)] for k in keys.tolist()])
    o = oracle.estep(im.pi, im.transition, keys, Etab, obs)
    ll = im.loglik()
    assert abs(ll - o["loglik"]) <= LL_TOL * abs(o["loglik"]), (ll, o["loglik"])
    assert rel_err(im.xisums[0], o["xisum"]) <= STAT_TOL
    for k, v in o["gamma_sums"].items():
        assert np.max(np.abs(im.gamma_sums[0][k] - v)) <= STAT_TOL * max(np.abs(v).max(), 1e-300), k
    assert rel_err(im.gammas[0][:, 0], o["gamma"][:, 0]) <= STAT_TOL
    q = np.array(im.Q(separate=True))
    assert np.all(np.abs(q - o["q"]) <= STAT_TOL * np.maximum(np.abs(o["q"]), 1e-12)), (q, o["q"])
    # one chunk = the sequential algorithm: the same numbers
    im1 = _manager(M, n, obs, chunk=10 ** 6)
    im1.E_step()
    assert abs(im1.loglik() - ll) <= 1e-9 * abs(ll)
    assert rel_err(im1.xisums[0], im.xisums[0]) <= STAT_TOL


def test_m512_vs_compiled_reference():
    from smcpp_amd import _smcpp, synth
    from smcpp_amd.model import PiecewiseModel
    z = np.load(os.path.join(GOLDEN, "G21_M512_n10_1000rows.npz"))
    g = {k: z[k] for k in z.files}
    n, rows = int(g["n"]), int(g["rows"])
    obs = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, n)[:rows], dtype=np.int32)
    assert synth.contig_crc(obs) == int(g["crc"])
    im = _smcpp.PyOnePopInferenceManager(n, [obs], g["hs"], ("pop1",), float(g["pol"]))
    im.model = PiecewiseModel(g["a"], g["s"], 1e4, "pop1")
    im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
    im.set_chunking(250)
    im.E_step()
    assert im.chain_mode() == 5
    # the engine's own preparation against the reference's at M = 512
    np.testing.assert_allclose(im.pi, g["pi"], rtol=1e-12)
    T = im.transition
    np.testing.assert_allclose(np.diag(T), g["T_diag"], rtol=1e-10)
    np.testing.assert_allclose(T.sum(axis=1), g["T_rowsum"], rtol=1e-12)
    ep = im.emission_probs
    ref_E = {tuple(int(x) for x in k): e for k, e in zip(g["keys"], g["E"])}
    for k in im.keys.tolist():
        np.testing.assert_allclose(ep[tuple(k)], ref_E[tuple(k)], rtol=1e-9, atol=1e-16)
    ll = im.loglik()
    assert abs(ll - float(g["loglik"])) <= LL_TOL * abs(float(g["loglik"])), (ll, float(g["loglik"]))
    xs = im.xisums[0]
    for got, want in ((xs.sum(axis=1), g["xisum_rowsum"]), (xs.sum(axis=0), g["xisum_colsum"]), (np.diag(xs), g["xisum_diag"])):
        assert np.max(np.abs(got - want)) <= STAT_TOL * np.abs(want).max()
    assert abs(xs.sum() - float(g["xisum_total"])) <= STAT_TOL * abs(float(g["xisum_total"]))
    keys = [tuple(int(x) for x in k) for k in g["keys"]]
    got = im.gamma_sums[0]
    for k, v, h in zip(keys, g["gs"], g["gs_have"]):
        if h:
            assert np.max(np.abs(got[k] - v)) <= STAT_TOL * max(np.abs(v).max(), 1e-300), k
    assert rel_err(im.gammas[0][:, 0], g["gamma0"]) <= STAT_TOL
    q = np.array(im.Q(separate=True))
    assert np.all(np.abs(q - g["q"]) <= STAT_TOL * np.maximum(np.abs(g["q"]), 1e-12)), (q, g["q"])


def test_beyond_256_states_unbuilt_paths_fail_loudly():
    from smcpp_amd import synth
    n = 10
    obs = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, n)[:500], dtype=np.int32)
    im = _manager(300, n, obs)
    im.save_gamma = True
    with pytest.raises(RuntimeError, match="256"):
        im.E_step()
    im = _manager(300, n, obs)
    im.E_step()
    rng = np.random.default_rng(0)
    T = rng.random((300, 300)); T /= T.sum(axis=1, keepdims=True)
    ep = im.emission_probs
    keys = im.keys
    im.set_raw(im.pi, T, keys, np.array([ep[tuple(k)] for k in keys.tolist()]))
    with pytest.raises(RuntimeError, match="256"):
        im.E_step()
    with pytest.raises(RuntimeError):
        _manager(600, n, obs)
