"""CPU tests of the checker itself: the plain-C restatement (oracle/) against the golden vectors emitted by the
compiled reference, and — in the build container, where oracle/_ref exists — against the compiled reference live."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import oracle, prep_oracle, ref


def test_oracle_matches_golden(golden):
    g = golden
    o = oracle.estep(g["pi"], g["T"], g["keys"], g["E"], g["obs"], save_gamma=True)
    assert abs(o["loglik"] - float(g["loglik"])) <= 1e-9 * abs(float(g["loglik"]))
    assert rel_err(o["xisum"], g["xisum"]) <= 5e-6
    ref_keys = [tuple(int(x) for x in k) for k in g["gs_keys"]]
    assert sorted(o["gamma_sums"].keys()) == sorted(ref_keys)
    for k, v in zip(ref_keys, g["gs_vals"]):
        assert np.max(np.abs(o["gamma_sums"][k] - v)) <= 5e-6 * max(np.abs(v).max(), 1e-300)
    assert np.all(np.abs(o["q"] - g["q"]) <= 1e-6 * np.maximum(np.abs(g["q"]), 1e-12))
    arg = o["gamma"].argmax(axis=0)
    strong = g["gamma_margin"] > 1e-5
    assert np.all((arg == g["gamma_argmax"]) | ~strong)
    st = int(g["alpha_stride"])
    a_sub = o["alpha_hat"][::st]
    assert np.max(np.abs(a_sub - g["alpha_sub"])) <= 1e-6
    assert np.max(np.abs(o["log_c"] - g["log_c"])) <= 1e-6


def test_g1_known_answer():
    """the number quoted there was produced by the full reference stack (with GSL)."""
    g = load_golden("G1_M16_n4")
    assert float(g["loglik"]) == -3108.781616833272
    np.testing.assert_allclose(g["q"], [-0.04690854380622089, -1692.7717048390532, -1369.8298321315315,
                                        -66.24468495398351], rtol=1e-12)
    assert abs(g["T"].sum(axis=1)[0] - (1 - 1e-5 / 17)) < 1e-12


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_oracle_vs_compiled_reference_live():
    from smcpp_amd import synth
    g = load_golden("G3_M32_n10_2Mbp")
    obs = synth.synth_contig(5, 600_000, 10)
    r = ref.estep(g["pi"], g["T"], g["keys"], g["E"], obs, save_gamma=True, want_alpha=True)
    o = oracle.estep(g["pi"], g["T"], g["keys"], g["E"], obs, save_gamma=True)
    assert abs(o["loglik"] - r["loglik"]) <= 1e-10 * abs(r["loglik"])
    assert (o["alpha_hat"] == r["alpha_hat"]).mean() > 0.99           # float alpha is bit-identical almost everywhere
    assert rel_err(o["xisum"], r["xisum"]) <= 1e-6
    assert np.array_equal(o["gamma"].argmax(0), r["gamma"].argmax(0))


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_emission_assembly_reproduces_golden_tables():
    """prep_oracle.emission_probs on the reference's CSFS must give the emission table stored in the fixtures."""
    g = load_golden("G4_M64_n20_2Mbp")
    p = ref.prep(g["a"], g["s"], g["hs"], float(g["rho"]), float(g["theta"]), int(g["n"]))
    ep = prep_oracle.emission_probs(g["keys"], int(g["n"]), p["csfs"], p["avg_ct"], float(g["theta"]),
                                    float(g["alpha"]), float(g["pol"]))
    E = np.array([ep[tuple(int(x) for x in k)] for k in g["keys"]])
    np.testing.assert_allclose(E, g["E"], rtol=1e-13)
    np.testing.assert_allclose(p["T"], g["T"], rtol=1e-13)
    np.testing.assert_allclose(p["pi"], g["pi"], rtol=1e-13)


def test_hypergeometric_pdf():
    from scipy.stats import hypergeom
    for (k, n1, n2, t) in [(0, 3, 5, 2), (2, 4, 4, 4), (1, 10, 2, 5), (3, 3, 7, 6)]:
        assert abs(prep_oracle.hypergeom_pdf(k, n1, n2, t) - hypergeom.pmf(k, n1 + n2, n1, t)) < 1e-12
