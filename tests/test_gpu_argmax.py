"""Posterior decode indices AT SCALE against the compiled reference (goldens G19 / G20, tests/golden/make_golden_argmax.py:
`HMM::Estep` with `save_gamma`, /root/reference/src/hmm.cpp:141-150, and the per-column argmax `smc++ posterior` reports,
/root/reference/smcpp/commands/posterior.py:98-111, generated in the build container):

  G19_headline / G19_c2          contig 0 of the headline (M = 64) and of config C2 (M = 32): 235 553 columns each
  G20_posterior / G20_posterior64  the 10^6-row un-binned contigs of `bench.py --workload posterior / posterior64`

Two routes per golden, both on the DEFAULT chain family:
  raw     `set_raw` with the parameters the reference ran on;
  params  `im.model = ...` (the route `bench.py` times: cold preparation on the device), where additionally the statistics,
          Q and the log-likelihood of the lean (no `save_gamma`) E-step are compared at full size.

Bar (BASELINE.json north_star): the decoded index is identical on EVERY column whose reference top-1 / top-2 relative margin
exceeds 1e-5; columns below that are counted and printed (the goldens hold every column with a margin below 1e-3).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

LL_TOL = 1e-6
STAT_TOL = 5e-6
CASES = ["G19_headline", "G19_c2", "G20_posterior", "G20_posterior64"]


def _load(name):
    from smcpp_amd import synth
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    if name.startswith("G19"):
        p = np.load(os.path.join(GOLDEN, "params_M64_n20.npz" if name == "G19_headline" else "params_M32_n10.npz"))
        for k in ("pi", "T", "E", "hs", "a", "s", "theta", "rho", "alpha", "pol", "n"):
            g[k] = p[k]
        assert np.array_equal(p["keys"], g["keys"])
        obs = synth.synth_contig(0, 100_000_000, int(g["n"]))
    else:
        obs = synth.synth_posterior_contig(1_000_000, int(g["n"]), seed=7)
    assert len(obs) == int(g["rows"]) and synth.contig_crc(obs) == int(g["crc"]), "the generator no longer produces the golden's rows"
    return g, np.ascontiguousarray(obs, dtype=np.int32)


def _manager(g, obs, route):
    from smcpp_amd import _smcpp
    from smcpp_amd.model import PiecewiseModel
    im = _smcpp.PyOnePopInferenceManager(int(g["n"]), [obs], g["hs"], ("pop1",), float(g["pol"]))
    im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
    if route == "raw":
        im.set_raw(g["pi"], g["T"], g["keys"], g["E"])
    else:
        im.model = PiecewiseModel(g["a"], g["s"], 1e4, "pop1")
    return im


def argmax_report(arg, g):
    """-> (mismatching columns, of which with a reference margin > 1e-5)"""
    mism = np.nonzero(np.asarray(arg).astype(np.int64) != g["gamma_argmax"].astype(np.int64))[0]
    margin = np.full(len(g["gamma_argmax"]), float(g["low_margin_below"]))
    margin[g["low_margin_cols"]] = g["low_margin"]
    return mism, mism[margin[mism] > 1e-5], margin


def _check_stats(im, g):
    ll = im.loglik()
    assert abs(ll - float(g["loglik"])) <= LL_TOL * abs(float(g["loglik"])), (ll, float(g["loglik"]))
    assert rel_err(im.xisums[0], g["xisum"]) <= STAT_TOL, rel_err(im.xisums[0], g["xisum"])
    got = im.gamma_sums[0]
    keys = [tuple(int(x) for x in k) for k in g["keys"]]
    assert sorted(got.keys()) == sorted(k for k, h in zip(keys, g["gs_have"]) if h)
    for k, v, h in zip(keys, g["gs"], g["gs_have"]):
        if h:
            assert np.max(np.abs(got[k] - v)) <= STAT_TOL * max(np.abs(v).max(), 1e-300), k
    q = np.array(im.Q(separate=True))
    assert np.all(np.abs(q - g["q"]) <= STAT_TOL * np.maximum(np.abs(g["q"]), 1e-12)), (q, g["q"])
    return ll


@pytest.mark.parametrize("route", ["raw", "params"])
@pytest.mark.parametrize("name", CASES)
def test_posterior_indices_at_scale_vs_compiled_reference(name, route):
    g, obs = _load(name)
    im = _manager(g, obs, route)
    if route == "params":
        # the lean E-step of the route bench.py times: statistics, Q and loglik at full size against the reference's
        im.E_step()
        ll = _check_stats(im, g)
        if name.startswith("G19"):
            assert im.chain_mode() == 5
        print(f"{name}[params] lean E-step: loglik rel {abs(ll - float(g['loglik'])) / abs(float(g['loglik'])):.2e}, "
              f"xisum rel {rel_err(im.xisums[0], g['xisum']):.2e}")
    im.save_gamma = True
    im.E_step()
    ll = im.loglik()
    assert abs(ll - float(g["loglik"])) <= LL_TOL * abs(float(g["loglik"])), (ll, float(g["loglik"]))
    arg = im.gamma_argmax(0)
    assert arg.shape == g["gamma_argmax"].shape
    mism, strong, margin = argmax_report(arg, g)
    print(f"{name}[{route}]: {len(arg)} columns, chain mode {im.chain_mode()}, argmax mismatches {len(mism)} "
          f"(reference margin > 1e-5: {len(strong)}); reference columns below 1e-5: {int((margin < 1e-5).sum())}, "
          f"smallest reference margin {float(g['min_margin']):.2e}"
          + (f"; mismatching columns {mism[:8].tolist()} margins {margin[mism][:8].tolist()}" if len(mism) else ""))
    assert len(strong) == 0, f"posterior argmax differs on columns {strong[:10]} whose reference margin exceeds 1e-5"
    # the per-column posterior itself on the golden's strided sample (and the device argmax against the host's on it)
    gam = im.gammas[0]
    st = int(g["gamma_stride"])
    sub = gam[:, ::st]
    assert sub.shape == g["gamma_sub"].shape
    assert np.max(np.abs(sub - g["gamma_sub"])) <= 2e-5 * max(1.0, float(np.abs(g["gamma_sub"]).max()))
    assert np.array_equal(gam.argmax(axis=0).astype(np.int64), np.asarray(arg).astype(np.int64))
    assert rel_err(gam[:, 0], g["gamma0"]) <= STAT_TOL


def test_repeated_save_gamma_steps_keep_every_column():
    """Round 6 regression: the per-row posterior buffer used to be cleared on the main stream BEHIND the scan chains' fork event, so the
    span-1 branch on its side stream could write rows the memset then wiped (zero columns on a timing-dependent 2 - 7 % of the
    headline contig).  Now only row 0 is cleared.  Twelve save_gamma E-steps on one manager, a lean E-step in between: every column of
    every step sums to its span and decodes the same index."""
    g, obs = _load("G19_headline")
    im = _manager(g, obs, "params")
    im.E_step()
    im.save_gamma = True
    spans = np.concatenate([[1.0], obs[:, 0].astype(float)])
    ref = None
    for it in range(12):
        if it == 6:
            im.save_gamma = False
            im.E_step()
            im.save_gamma = True
        im.E_step()
        arg = np.asarray(im.gamma_argmax(0)).astype(np.int64)
        if ref is None:
            ref = arg
            mism, strong, _ = argmax_report(arg, g)
            assert len(strong) == 0
        assert np.array_equal(arg, ref), f"step {it}: {int((arg != ref).sum())} columns changed"
        if it in (0, 7, 11):
            gam = im.gammas[0]
            np.testing.assert_allclose(gam.sum(axis=0)[1:], spans[1:], rtol=1e-9)      # (column 0 is alpha_0 o beta_0 as it stands, hmm.cpp:150)


@pytest.mark.parametrize("name", ["G19_headline", "G19_c2"])
def test_posterior_indices_with_float_scans_in_the_stored_passes(engine_opt, name):
    """the timed path runs every scan of the stored passes in float (chains_ss.hpp: ss_x_scan_fwd / _bwd); until
    round 6 `save_gamma` switched those float scans off, so the path whose indices were compared was not the path that was timed.
    Now `save_gamma` keeps them (the default): on the full-size binned contigs the decoded index must equal the compiled reference's
    on every column whose reference margin exceeds 1e-5 (the north-star bar as SURVEY.md 8(c) states it) - measured: on EVERY
    column; the fp64 scans (SMCPP_SS_MIXED=0) are run beside them and must decode the same indices."""
    g, obs = _load(name)
    im = _manager(g, obs, "params")
    im.save_gamma = True
    im.E_step()
    assert im.chain_mode() == 5 and im.describe()["plan"]["float_scans_in_stored_passes"]
    ll = im.loglik()
    assert abs(ll - float(g["loglik"])) <= LL_TOL * abs(float(g["loglik"]))
    arg = im.gamma_argmax(0)
    mism, strong, margin = argmax_report(arg, g)
    print(f"{name}[float scans + save_gamma]: {len(arg)} columns, argmax mismatches {len(mism)} (reference margin > 1e-5: {len(strong)})"
          + (f"; mismatching columns {mism[:8].tolist()} margins {margin[mism][:8].tolist()}" if len(mism) else ""))
    assert len(strong) == 0
    gam = im.gammas[0]
    st = int(g["gamma_stride"])
    assert np.max(np.abs(gam[:, ::st] - g["gamma_sub"])) <= 2e-5 * max(1.0, float(np.abs(g["gamma_sub"]).max()))
    engine_opt("SMCPP_SS_MIXED", "0")
    im.E_step()
    assert not im.describe()["plan"]["float_scans_in_stored_passes"]
    arg64 = im.gamma_argmax(0)
    mism64, strong64, _ = argmax_report(arg64, g)
    print(f"{name}[fp64 scans + save_gamma]: argmax mismatches {len(mism64)}; float-scan vs fp64-scan decode differs on "
          f"{int(np.count_nonzero(np.asarray(arg) != np.asarray(arg64)))} columns")
    assert len(strong64) == 0
