"""The callers of the path that make up `smc++ estimate`: SMCModel, the two-stage Analysis, the EM
loop with the log-likelihood monitor and the model.final.json dump (smcpp/analysis/*.py, smcpp/optimize/*)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT


def test_smcmodel_pieces_seeds_and_dict():
    from smcpp_amd.analysis import SMCModel, PIECES
    knots = np.array([0.01, 0.05, 0.2, 1.0, 4.0])
    m = SMCModel(knots, 1e4, "pop1")
    m[:] = np.log([2.0, 0.5, 1.5, 3.0, 0.8])
    s = m.s
    assert len(s) == PIECES and s[0] == knots[0] and abs(s.sum() - knots[-1]) < 1e-12      # smcpp/model.py:117-128
    a = m.stepwise_values()
    # piecewise constant: the value of the knot interval containing the right end of each piece, flat outside the knots
    t = np.cumsum(s)
    np.testing.assert_allclose(a[t < knots[1] * (1 - 1e-12)], 2.0)
    np.testing.assert_allclose(a[-1], 0.8)
    assert abs(m.regularizer() - (np.diff(np.log([2.0, 0.5, 1.5, 3.0, 0.8]), 2) ** 2).sum()) < 1e-14
    # seeds = d a_k / d y_j, against finite differences; clipped pieces have zero derivative
    m.differentiate([1, 3, 4])
    S = m.derivative_seeds()
    y0 = np.array(m[:], dtype=float)
    for j, c in enumerate([1, 3, 4]):
        h = 1e-6
        m[c] = y0[c] + h; ap = m.stepwise_values()
        m[c] = y0[c] - h; am = m.stepwise_values()
        m[c] = y0[c]
        np.testing.assert_allclose(S[:, j], (ap - am) / (2 * h), rtol=1e-6, atol=1e-9)
    g = m.regularizer_gradient()
    for c in range(5):
        h = 1e-6
        m[c] = y0[c] + h; rp = m.regularizer()
        m[c] = y0[c] - h; rm = m.regularizer()
        m[c] = y0[c]
        assert abs(g[c] - (rp - rm) / (2 * h)) < 1e-6
    m[0] = np.log(1e6)
    m.differentiate([0])
    assert np.all(m.derivative_seeds()[m.stepwise_values() >= 1e3] == 0)
    d = m.to_dict()
    assert d["class"] == "SMCModel" and d["spline_class"] == "Piecewise" and d["pid"] == "pop1"
    m2 = SMCModel.from_dict(json.loads(json.dumps(d)))
    np.testing.assert_array_equal(m2.stepwise_values(), m.stepwise_values())


def test_model_classes_match_the_reference_python_classes():
    """Golden G15 (tests/golden/make_golden_model.py): the reference's OWN smcpp/model.py + smcpp/spline/piecewise.py and the knot
    placement / regularisation scaling of its Analysis (analysis.py:104-126), imported where they lie and run on fixed inputs:
    `s`, `stepwise_values` (with the clipping to [1e-3, 1e3]), `__call__`, `regularizer`, `to_dict` / `from_dict`, `randomize`
    under a fixed seed, `_init_knots`, `_init_regularization`."""
    import types
    from conftest import load_golden
    from smcpp_amd.analysis import Analysis, SMCModel
    g = load_golden("G15_model")
    for ci in range(int(g["n_cases"])):
        m = SMCModel(g[f"c{ci}_knots"], 1e4, "pop1")
        m[:] = g[f"c{ci}_y"]
        np.testing.assert_allclose(m.s, g[f"c{ci}_s"], rtol=1e-14)
        np.testing.assert_allclose(m.stepwise_values(), g[f"c{ci}_stepwise"], rtol=1e-14)
        np.testing.assert_allclose(m(g[f"c{ci}_points"]), g[f"c{ci}_values"], rtol=1e-14)
        assert abs(m.regularizer() - float(g[f"c{ci}_regularizer"])) <= 1e-12 * max(1.0, abs(float(g[f"c{ci}_regularizer"])))
        ref_d = json.loads(str(g[f"c{ci}_dict"]))
        d = m.to_dict()
        assert sorted(d) == sorted(ref_d)
        for k in ("class", "spline_class", "pid", "N0"):
            assert d[k] == ref_d[k]
        np.testing.assert_allclose(d["knots"], ref_d["knots"], rtol=0)
        np.testing.assert_allclose(d["y"], ref_d["y"], rtol=0)
        m2 = SMCModel.from_dict(ref_d)                       # a file the reference wrote is read back
        np.testing.assert_array_equal(m2.stepwise_values(), m.stepwise_values())
        np.random.seed(3)
        m.randomize()
        np.testing.assert_allclose(m[:], g[f"c{ci}_randomized"], rtol=1e-15)
    for k in range(int(g["n_knot_cases"])):
        t1, tK = [None if np.isnan(x) else float(x) for x in g[f"k{k}_t"]]
        ns = types.SimpleNamespace()
        Analysis._init_knots(ns, g[f"k{k}_hs"], t1, tK)
        np.testing.assert_allclose(ns._knots, g[f"k{k}_knots"], rtol=1e-15)
    for q, rp, lam, pen in g["penalty_cases"]:
        ns = types.SimpleNamespace(_args=types.SimpleNamespace(lambda_=None if np.isnan(lam) else float(lam),
                                                               regularization_penalty=float(rp)), Q=lambda q=q: float(q))
        Analysis._init_regularization(ns)
        assert abs(ns._penalty - pen) <= 1e-14 * abs(pen)


def test_two_population_piece_cuts_are_kept_between_updates():
    """`TwoPopulationModel.for_pop(pop 2)` keeps what depends on the piece LENGTHS and the split only (where the pieces are cut,
    which piece serves which interval) between updates - round 5: 70 -> 6 us of numpy per `model =`: the kept structure must give
    exactly what a fresh derivation gives after the sizes, the split or the lengths changed."""
    from smcpp_amd.model import PiecewiseModel, TwoPopulationModel
    rng = np.random.default_rng(5)
    s1 = np.array([0.01, 0.02, 0.05, 0.1, 0.3, 0.7, 1.5, 1.0])
    s2 = np.array([0.03, 0.07, 0.2, 0.4, 1.0])
    for split in (1e-3, 0.013, 0.5, 0.38, 3.7, 200.0):              # inside the first piece ... beyond the last knot
        m1 = PiecewiseModel(rng.uniform(0.5, 3.0, len(s1)), s1, 1e4, pid="pop1")
        m2 = PiecewiseModel(rng.uniform(0.5, 3.0, len(s2)), s2, 1e4, pid="pop2")
        model = TwoPopulationModel(m1, m2, split)

        def fresh():
            return TwoPopulationModel(PiecewiseModel(m1.a.copy(), m1.s.copy(), 1e4, pid="pop1"),
                                      PiecewiseModel(m2.a.copy(), m2.s.copy(), 1e4, pid="pop2"), model.split).for_pop("pop2")

        def same(p, q):
            assert np.array_equal(p.stepwise_values(), q.stepwise_values()) and np.array_equal(p.s, q.s)
        same(model.for_pop("pop2"), fresh())
        m2[1] = 7.0; m1[0] = 0.3; m1[len(s1) - 1] = 2.2             # sizes move: the structure is reused
        same(model.for_pop("pop2"), fresh())
        model.split = split * 1.7                                    # the split moves: re-derived
        same(model.for_pop("pop2"), fresh())
        same(model.for_pop("pop1"), m1)


def _example_contig():
    from smcpp_amd import vcf2smc as V
    c, _ = V.vcf2smc(os.path.join(ROOT, "tests", "golden", "example.vcf.gz"), "1", ("pop1", ["msp_0", "msp_1", "msp_2"]))
    return c


def test_model_final_json_field_by_field_against_the_reference(tmp_path):
    """Golden G17 (tests/golden/make_golden_final_json.py): `model.final.json` written by the REFERENCE's own `BaseAnalysis.dump`
    for its own `SMCModel` at a fixed parameter point of the example run - hidden states from its `balance_hidden_states` (rate
    function served by the compiled reference), knots from its `Analysis._init_knots`.  This repository's classes, fed the same
    inputs, must write the same file: same keys, same strings, every number to 1e-12 (the hidden states come out of Brent
    root-finding on both sides), `Infinity` where the reference writes it."""
    import types
    from smcpp_amd import analysis as A
    from smcpp_amd.posterior import balance_hidden_states
    gin = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G17_inputs.npz"))
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G17_model_final.json")))
    N0, mu, w, M = float(gin["N0"]), float(gin["mu"]), int(gin["w"]), int(gin["M"])
    m0 = A.SMCModel(gin["knots0"], N0, "pop1")
    m0[:] = gin["y"]
    hs = balance_hidden_states(m0, M)                      # coalescent units (the reference's, in generations, / 2 N0)
    ns = types.SimpleNamespace()
    A.Analysis._init_knots(ns, hs, None, None)
    m = A.SMCModel(ns._knots, N0, "pop1")
    m[:] = gin["yf"]
    A.write_final_json(str(tmp_path / "model.final"), 2.0 * N0 * mu, 2.0 * N0 * mu, w, m, {"pop1": hs})
    txt = open(tmp_path / "model.final.json").read()
    got = json.loads(txt)

    def same(a, b, path):
        assert type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float))), (path, a, b)
        if isinstance(a, dict):
            assert sorted(a) == sorted(b), path
            for k in a:
                same(a[k], b[k], path + "/" + k)
        elif isinstance(a, list):
            assert len(a) == len(b), path
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, f"{path}[{i}]")
        elif isinstance(a, float):
            assert (np.isinf(a) and np.isinf(b)) or abs(a - b) <= 1e-12 * max(abs(b), 1e-300), (path, a, b)
        else:
            assert a == b, (path, a, b)
    same(got, ref, "")
    assert "Infinity" in txt and txt.startswith("{\n    \"alpha\": 100,")          # sort_keys, indent = 4, as json.dump writes them


@pytest.mark.gpu
def test_analysis_replays_the_reference_run_on_the_example(tmp_path):
    """`np.random.seed(0); Analysis([example, n = 4], knots=8, unfold=True, w=100, mu=1.25e-8,
    em_iterations=1, algorithm='L-BFGS-B', multi=True, regularization_penalty=6, spline='piecewise')` as the survey ran
    it on the reference (Python oracle): the bootstrap E-step sees 1 850 un-binned rows with ONE hidden state, the main
    E-step 2 727 rows with M = 15 balanced states (the mixture model needs more windows than a 1 Mbp contig has);
    their log-likelihoods were -9628.299008799679 and -1350.7839372364251."""
    from smcpp_amd.analysis import Analysis, EstimateArgs
    np.random.seed(0)
    args = EstimateArgs(knots=8, unfold=True, w=100, mu=1.25e-8, em_iterations=1, algorithm="L-BFGS-B", multi=True,
                        regularization_penalty=6, outdir=str(tmp_path), base="model")
    an = Analysis([_example_contig()], args)
    assert an.hidden_state_source == "balanced"
    assert len(an.hidden_states) == 16 and an.hidden_states[0] == 0 and np.isinf(an.hidden_states[-1])
    assert [len(c.data) for c in an.contigs] == [2727]
    assert abs(an.bootstrap_loglik - (-9628.299008799679)) <= 1e-6 * 9628.3
    an.run()
    ll = an._optimizer.logliks
    assert len(ll) == 1
    # the main model starts from the bootstrap M-step's optimum: four log-sizes fitted to 1 Mbp (ScaleOptimizer's common shift,
    # then L-BFGS-B inside +-3 log-unit bounds), some of them barely identified, so where the optimiser stops depends on its
    # path (this repository's gradients come from the engine, the reference's from its ad numbers).  Observed: -1351.0197
    # against the reference's -1350.7839 (1.7e-4 relative; 3.3e-4 before the scale step was reproduced) - the replay pins
    # the flow; the numbers of this data set at a FIXED parameter point are pinned below against the C restatement
    assert abs(ll[0] - (-1350.7839372364251)) <= 4e-4 * 1350.8, ll
    # ---- the engine on the main-stage data (2 727 thinned / binned rows, 15 balanced states) at a parameter point no
    # optimiser has touched: constant size at Watterson's estimate; against oracle/ (hmm.cpp restated) on the same inputs ----
    from oracle import oracle
    from smcpp_amd import _smcpp
    from smcpp_amd.analysis import SMCModel
    fixed = SMCModel(an.model.knots, an.model.N0, "pop1")
    fixed[:] = np.log(an._watterson / (2.0 * args.mu * an._N0))
    obs = [np.ascontiguousarray(c.data, dtype=np.int32) for c in an.contigs]
    im = _smcpp.PyOnePopInferenceManager(an.contigs[0].n[0], obs, an.hidden_states, ("pop1",), 0.0)
    im.theta = an._theta; im.rho = an._rho; im.alpha = args.w
    im.model = fixed
    im.E_step()
    ep = im.emission_probs
    keys = im.keys
    o = oracle.estep(im.pi, im.transition, keys, np.array([ep[tuple(k)] for k in keys.tolist()]), obs[0])
    assert abs(im.loglik() - o["loglik"]) <= 1e-6 * abs(o["loglik"])
    assert np.max(np.abs(im.xisums[0] - o["xisum"]) / np.abs(o["xisum"])) <= 5e-6
    assert np.all(np.abs(np.array(im.Q(separate=True)) - o["q"]) <= 5e-6 * np.maximum(np.abs(o["q"]), 1e-12))
    j = json.load(open(tmp_path / "model.final.json"))
    assert sorted(j) == ["alpha", "hidden_states", "model", "rho", "theta"]
    assert j["theta"] == 1e-4 and j["alpha"] == 100 and j["model"]["class"] == "SMCModel"
    assert len(j["hidden_states"]["pop1"]) == 16 and len(j["model"]["y"]) == len(j["model"]["knots"])
    assert os.path.exists(tmp_path / ".model.iter0.json")


@pytest.mark.gpu
def test_em_loop_terminates_on_the_loglik_monitor(tmp_path):
    """LoglikelihoodMonitor (plugins/loglikelihood_monitor.py): EM stops as soon as the relative improvement of the
    log-likelihood falls below ftol; until then it must not decrease (beyond the float-alpha noise)."""
    from smcpp_amd.analysis import Analysis, EstimateArgs
    np.random.seed(1)
    args = EstimateArgs(knots=6, unfold=True, w=100, em_iterations=12, multi=True, ftol=2e-3, r=1.25e-8)
    an = Analysis([_example_contig()], args)
    an.run()
    ll = np.array(an._optimizer.logliks)
    assert 2 <= len(ll) < 12, ll
    assert np.all(np.diff(ll) >= -1e-5 * np.abs(ll[:-1])), ll
    assert (ll[-2] - ll[-1]) / ll[-2] < 2e-3


@pytest.mark.gpu
def test_em_with_more_than_64_hidden_states_on_the_example(tmp_path):
    """`estimate` with 40 knots (79 hidden states: two states per lane, the halo pass, eigen-free statistics) on the example contig;
    three EM iterations must not lower the log-likelihood (beyond the float-alpha noise) and the model file must come out.  (The
    pipeline's own rows stay below 64 positions here: the cut rows of round 6 are exercised by tests/test_gpu_bigm.py.)"""
    from smcpp_amd.analysis import Analysis, EstimateArgs
    np.random.seed(2)
    args = EstimateArgs(knots=40, unfold=True, w=100, em_iterations=3, multi=True, ftol=1e-9, r=1.25e-8, outdir=str(tmp_path))
    an = Analysis([_example_contig()], args)
    M = len(an.hidden_states) - 1
    assert M > 64
    an.run()
    plan = an._im.describe()["plan"]
    print(plan)
    assert plan["eigen_free_statistics"] and plan["states"] == M and plan["max_span"] <= 64, plan
    ll = np.array(an._optimizer.logliks)
    assert len(ll) >= 3 and np.all(np.isfinite(ll)), ll
    assert np.all(np.diff(ll) >= -1e-5 * np.abs(ll[:-1])), ll
    import os
    assert os.path.exists(os.path.join(str(tmp_path), "model.final.json"))
