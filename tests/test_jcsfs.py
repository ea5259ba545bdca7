"""Two-population cold preparation (second half; rows A6-A8 for P = 2) on the host.

What pins it (the reference's `src/jcsfs.cpp` needs GSL headers and cannot be built in this image):
  * G9: the reference's pure-Python original `smcpp/jcsfs.py` — of which jcsfs.cpp is a line-by-line translation —
    executed from /root/reference with its calls into the compiled binding served by the compiled reference C++
    (`tests/golden/make_golden_jcsfs.py`), distinguished pair in population 1;
  * the invariants asserted by the reference's own tests (test/unit/test_jcsfs.py: marginal over population 2 equals
    the one-population CSFS, marginal over population 1 equals the folded SFS of population 2's lineage);
  * for one distinguished lineage per population (no Python original exists): both marginals of the implied joint SFS;
  * the emission-table assembly for 6-int keys against a literal Python restatement of the generic-P templates.
"""
import os

import numpy as np
import pytest

from oracle import prep_oracle, ref
from smcpp_amd import _engine as E, _smcpp
from smcpp_amd.model import PiecewiseModel, TwoPopulationModel

HERE = os.path.dirname(os.path.abspath(__file__))


def _models():
    m1 = PiecewiseModel([1.0, 4.0], [0.5, 1.0], 1e4, pid="pop1")
    m2 = PiecewiseModel([2.0, 4.0, 2.0], [0.1, 0.2, 0.3], 1e4, pid="pop2")
    return m1, m2


def _undist(sfs):
    a, n = sfs.shape[0] - 1, sfs.shape[1] - 1
    u = np.zeros(n + a)
    for i in range(a + 1):
        for j in range(n + 1):
            if i + j < n + a:
                u[i + j] += sfs[i, j]
    return u


@pytest.mark.parametrize("name", list("ABCDE"))
def test_together_matches_reference_python_original(name):
    g = np.load(os.path.join(HERE, "golden", "G9_jcsfs_together.npz"))
    n1, n2, K = (int(x) for x in g[name + "_n"])
    for si, sp in enumerate(g[name + "_splits"]):
        got = E.host_joint_csfs(n1, n2, 2, 0, g[name + "_hs"], (g[name + "_a1"], g[name + "_s1"]),
                                (g[name + "_a2"], g[name + "_s2"]), float(sp), K=K)
        want = g[name + "_J"][si]
        # the Python original clips at 0, the C++ translation floors at 1e-20 (jcsfs.cpp:229-238): compare above that
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-13)


def test_marginal_over_population_2_is_the_one_population_csfs():
    """test/unit/test_jcsfs.py:77-88; exact above the split, Monte-Carlo/truncation limited below it (the reference
    asserts 10 %)."""
    m1, m2 = _models()
    ts = [0.0, 1.0, 2.0, np.inf]
    n1, n2 = 5, 10
    for split in [0.1, 0.5, 1.0, 1.5, 2.5]:
        jc = E.host_joint_csfs(n1, n2, 2, 0, ts, (m1.a, m1.s), (m2.a, m2.s), split, K=200)
        for t1, t2, j in zip(ts[:-1], ts[1:], jc):
            A1 = _smcpp.raw_sfs(m1, n1, t1, t2)
            A2 = j.sum(axis=(-1, -2))
            tol = 1e-6 if split <= t1 else 1e-1
            assert np.allclose(A1.flat[1:-1], A2.flat[1:-1], rtol=tol, atol=0)


def test_marginal_over_population_1_is_population_2s_sfs():
    """test/unit/test_jcsfs.py:90-101 (asserted there to 10 %; holds to 1e-8)."""
    m1, m2 = _models()
    n1, n2 = 8, 10
    for split in [0.1, 0.25, 0.5, 0.75, 1.0, 2.0]:
        full2 = TwoPopulationModel(m1, m2, split).for_pop("pop2")
        A1 = _undist(_smcpp.raw_sfs(full2, n2 - 2, 0.0, np.inf))[1:]
        jc = E.host_joint_csfs(n1, n2, 2, 0, [0.0, np.inf], (m1.a, m1.s), (m2.a, m2.s), split, K=10)[0]
        A2 = jc.sum(axis=(0, 1, 2))[1:-1]
        np.testing.assert_allclose(A2, A1, rtol=1e-7)


@pytest.mark.parametrize("split", [0.02, 0.1, 0.5, 1.2])
def test_apart_marginals(split):
    """One distinguished lineage per population: folding (a_p, b_p) -> a_p + b_p gives the joint SFS of n1+1 and n2+1
    lineages; each marginal must be the ordinary SFS of that population's lineage history."""
    m1, m2 = _models()
    n1, n2 = 6, 7
    jc = E.host_joint_csfs(n1, n2, 1, 1, [0.0, np.inf], (m1.a, m1.s), (m2.a, m2.s), split, K=10)[0]
    jm = np.zeros((n1 + 2, n2 + 2))
    for a1 in range(2):
        for b1 in range(n1 + 1):
            for a2 in range(2):
                for b2 in range(n2 + 1):
                    jm[a1 + b1, a2 + b2] += jc[a1, b1, a2, b2]
    ref1 = _undist(_smcpp.raw_sfs(m1, n1 - 1, 0.0, np.inf))[1:]
    ref2 = _undist(_smcpp.raw_sfs(TwoPopulationModel(m1, m2, split).for_pop("pop2"), n2 - 1, 0.0, np.inf))[1:]
    np.testing.assert_allclose(jm.sum(axis=1)[1:n1 + 1], ref1, rtol=1e-7)
    np.testing.assert_allclose(jm.sum(axis=0)[1:n2 + 1], ref2, rtol=1e-7)
    # no coalescence of the distinguished pair before the split: states entirely below it stay empty
    hs = [0.0, split / 2, split, 2.0, np.inf]
    J = E.host_joint_csfs(n1, n2, 1, 1, hs, (m1.a, m1.s), (m2.a, m2.s), split, K=10)
    assert J[2].sum() > 1.0 and J[3].sum() > 1.0


@pytest.mark.parametrize("a1,a2", [(2, 0), (1, 1)])
def test_jacobian_vs_finite_differences(a1, a2):
    """test/unit/test_jcsfs.py:55-74 prints AD against differences for the C++ JointCSFS; asserted here."""
    m1, m2 = _models()
    n1, n2 = 6, 4
    hs = [0.0, 0.3, 1.0, np.inf]
    split = 0.45
    da1 = np.hstack([np.eye(2), np.zeros((2, 3))])
    da2 = np.hstack([np.zeros((3, 2)), np.eye(3)])
    J, dJ = E.host_joint_csfs(n1, n2, a1, a2, hs, (m1.a, m1.s), (m2.a, m2.s), split, K=10, da1=da1, da2=da2)
    np.testing.assert_allclose(J, E.host_joint_csfs(n1, n2, a1, a2, hs, (m1.a, m1.s), (m2.a, m2.s), split, K=10),
                               rtol=1e-11, atol=1e-15)
    h = 1e-6
    for k in range(5):
        ap, am = [m1.a.copy(), m2.a.copy()], [m1.a.copy(), m2.a.copy()]
        ap[k >= 2][k - 2 * (k >= 2)] += h
        am[k >= 2][k - 2 * (k >= 2)] -= h
        Jp = E.host_joint_csfs(n1, n2, a1, a2, hs, (ap[0], m1.s), (ap[1], m2.s), split, K=10)
        Jm = E.host_joint_csfs(n1, n2, a1, a2, hs, (am[0], m1.s), (am[1], m2.s), split, K=10)
        fd = (Jp - Jm) / (2 * h)
        assert np.abs(fd - dJ[..., k]).max() <= 5e-5 * max(1e-6, np.abs(dJ[..., k]).max())


@pytest.mark.parametrize("a1,a2", [(2, 0), (1, 1)])
def test_two_population_emission_table(a1, a2):
    """E from the engine's two-population preparation vs the literal Python restatement of the generic-P assembly fed
    with the same joint CSFS; pi / T must be the one-population quantities of the distinguished model."""
    m1, m2 = _models()
    n1, n2 = 4, 3
    hs = np.array([0.0, 0.2, 0.6, 1.5, np.inf])
    split, theta, rho, alpha, pol = 0.4, 1e-2, 2e-3, 1.0, 0.3
    tm = TwoPopulationModel(m1, m2, split)
    dist = tm.for_pop(None if a1 == 1 else "pop1")
    p1, p2 = tm.for_pop("pop1"), tm.for_pop("pop2")
    keys = []
    for A1 in ([-1] + list(range(a1 + 1))):
        for A2 in ([-1, 0] if a2 == 0 else [-1, 0, 1]):
            for nb1, b1 in [(0, 0), (2, 0), (2, 1), (4, 3), (4, 4)]:
                for nb2, b2 in [(0, 0), (1, 1), (3, 0), (3, 2)]:
                    keys.append((A1, b1, nb1, A2, b2, nb2))
    # keys whose only compatible configurations are non-segregating have no emission (construct_bins throws s<=0)
    ok = []
    for k in keys:
        try:
            prep_oracle.construct_bins_npop([k], (n1, n2), (a1, a2), pol)
            ok.append(k)
        except RuntimeError:
            pass
    keys = np.array(sorted(set(ok)), dtype=np.int32)
    assert len(keys) > 100
    pi, T, Etab = E.host_prep_twopop(n1, n2, a1, a2, hs, pol, (dist.a, dist.s), (p1.a, p1.s), (p2.a, p2.s), split,
                                     theta, rho, alpha, keys)
    J = E.host_joint_csfs(n1, n2, a1, a2, hs, (p1.a, p1.s), (p2.a, p2.s), split, K=10)
    tens = prep_oracle.incorporate_theta(J, theta)
    _, ct = E.host_rate_function(dist.a, dist.s, [0.0], hs)
    want = prep_oracle.emission_probs_npop([tuple(k) for k in keys], (n1, n2), (a1, a2), tens, ct, theta, alpha, pol)
    for k, e in zip(keys, Etab):
        np.testing.assert_allclose(e, want[tuple(int(x) for x in k)], rtol=1e-12)
    if a1 == 2:
        okeys = np.array([[0, 0, 0]], dtype=np.int32)
        pi1, T1, _ = E.host_prep_onepop(n1, hs, pol, dist.a, dist.s, theta, rho, alpha, okeys)
        np.testing.assert_array_equal(pi, pi1)
        np.testing.assert_array_equal(T, T1)
    else:
        assert np.isnan(ct[0]) and pi[0] == pytest.approx(1e-20 / (1 + 1e-20), rel=1e-6) or pi[0] < 1e-15
    assert np.all(Etab > 0) and np.all(Etab <= 1)


# ---- a1 = a2 = 1 ("apart", src/jcsfs.cpp:258-367): golden G12 ------------------------------------------------------
# jcsfs.cpp cannot be compiled here (GSL) and the reference's Python original only covers a1 = 2, so G12 is assembled
# by oracle/jcsfs_apart_oracle.py from the COMPILED reference building blocks the C++ calls (shiftParams /
# truncateParams, OnePopConditionedSFS::compute, R, modified_moran_rate_matrix); only the assembly loops, expm and the
# hypergeometric weights are restated there (in numpy / scipy, independently of the product's C++).
def _g12():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G12_jcsfs_apart.npz"))


def test_apart_matches_golden_from_compiled_reference_pieces():
    from smcpp_amd import _engine
    g = _g12()
    a1, a2, s = g["a1"], g["a2"], g["s"]
    for i in range(int(g["ncases"])):
        n1, n2 = (int(x) for x in g[f"c{i}_n"])
        J = _engine.host_joint_csfs(n1, n2, 1, 1, g[f"c{i}_hs"], (a1, s), (a2, s), float(g[f"c{i}_split"]))
        R = g[f"c{i}_J"]
        # entries are times in coalescent units (up to O(1)); 5e-15 absolute = a few ulp of the largest ones
        np.testing.assert_allclose(J, R, rtol=1e-10, atol=5e-15, err_msg=f"case {i}: n=({n1},{n2})")
        assert np.all(J[:, 0, 0, 0, 0] == 0) and np.all(J[:, 1, n1, 1, n2] == 0)


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_apart_matches_live_oracle_on_other_models():
    from oracle import jcsfs_apart_oracle as JA
    from smcpp_amd import _engine
    rng = np.random.default_rng(5)
    for n1, n2 in ((2, 7), (5, 1)):
        a1 = np.exp(rng.normal(0, 0.7, 5)); a2 = np.exp(rng.normal(0, 0.7, 3))
        s1 = np.exp(rng.normal(-2, 0.8, 5)); s2 = np.exp(rng.normal(-2, 0.8, 3))
        hs = np.r_[0.0, np.sort(np.exp(rng.normal(-1.5, 1.2, 7))), np.inf]
        for split in (0.0, float(hs[2]), 0.33):
            J = _engine.host_joint_csfs(n1, n2, 1, 1, hs, (a1, s1), (a2, s2), split)
            R = JA.joint_csfs_apart(n1, n2, hs, (a1, s1), (a2, s2), split)
            np.testing.assert_allclose(J, R, rtol=1e-10, atol=5e-15)


def test_c4_real_shape_parameters_match_reference_fixture():
    """Config C4 at its real shape (M = 48, n1 = n2 = 10, a = (2, 0), split 0.5; golden G13 = the reference's Python
    JointCSFS backed by the compiled C++ + ref_prep + the literal emission templates): the engine's two-population
    cold preparation, for every key of the C4 contig."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G13_c4_params.npz"))
    m1, m2 = (g["a1"], g["s1"]), (g["a2"], g["s2"])
    pi, T, Etab = E.host_prep_twopop(10, 10, 2, 0, g["hs"], float(g["pol"]), m1, m1, m2, float(g["split"]),
                                     float(g["theta"]), float(g["rho"]), float(g["alpha"]), g["keys"])
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-13)
    np.testing.assert_allclose(T, g["T"], rtol=1e-11, atol=1e-17)
    assert len(g["keys"]) >= 150
    np.testing.assert_allclose(Etab, g["E"], rtol=1e-8, atol=1e-14)
    J = E.host_joint_csfs(10, 10, 2, 0, g["hs"], m1, m2, float(g["split"]))
    np.testing.assert_allclose(J, g["J"], rtol=1e-9, atol=2e-14)
