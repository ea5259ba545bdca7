"""Scan chains on the semiseparable structure of the transition matrix (smcpp_amd/csrc/chains_ss.hpp).

The reference builds T in HJTransition (src/transition.cpp:176-254): constant columns below the diagonal, a rank-one part plus
one constant above it.  These tests check, on the GPU,
  * one position of both chains (the Kogge-Stone scans over the lanes) against the dense products with the golden T's of the
    compiled reference, for 1 .. 4 states per lane (M = 16 .. 256);
  * that a matrix without that structure is refused (the engine then runs the dense kernels);
  * the whole E-step on the scan kernels against the goldens is covered by tests/test_gpu_parity.py (chain family "default").
"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def ss_apply(T, x, e, float_scans=False):
    from smcpp_amd import _engine
    L = _engine.lib()
    T = np.ascontiguousarray(T, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    of = np.empty_like(x)
    ob = np.empty_like(x)
    fn = L.smcpp_debug_ss_apply_float_scans if float_scans else L.smcpp_debug_ss_apply
    rc = fn(T.shape[0], _engine.dptr(T), x.shape[0], _engine.dptr(x), _engine.dptr(e), _engine.dptr(of), _engine.dptr(ob))
    if rc == 1:
        raise RuntimeError(L.smcpp_last_error().decode())
    return rc, of, ob


def check_products(of, ob, ref_f, ref_b):
    """Sums over "the other side" of a state are total - prefix: an entry 1e-12 below the largest of its vector carries
    that cancellation (1e-16 of the TOTAL), far below the 1e-10 floor the reference puts under alpha; so the bound is
    1e-11 of the vector's largest entry (observed 1e-12 at M = 256: 8 scan levels over 4 states per lane) and 1e-7 per entry."""
    for o, r in ((of, ref_f), (ob, ref_b)):
        err = np.abs(o - r)
        assert np.all(np.isfinite(o))
        assert np.max(err / np.max(np.abs(r), axis=1, keepdims=True)) < 1e-11
        assert np.max(err / np.abs(r)) < 1e-7


@pytest.mark.parametrize("name", ["G1_M16_n4", "G3_M32_n10_2Mbp", "G5_M48_twopop_layout", "G4_M64_n20_2Mbp",
                                  "params_M256_n50"])
def test_one_position_matches_dense_products(name):
    g = load_golden(name)
    T = np.asarray(g["T"], dtype=np.float64)
    M = T.shape[0]
    rng = np.random.default_rng(7)
    nvec = 37
    # vectors with the dynamic range of real alpha / beta rows (entries down to 1e-12 of the largest)
    x = rng.random((nvec, M)) ** 8 + 1e-12
    x[0] = 1.0
    x[1] = 0.0; x[1, M - 1] = 1.0
    x[2] = 0.0; x[2, 0] = 1.0
    e = 0.2 + 0.8 * rng.random((nvec, M))
    ref_f = e * (x @ T)               # e o (T^T x)
    ref_b = (e * x) @ T.T             # T (e o x)
    rc, of, ob = ss_apply(T, x, e)
    assert rc == 0
    check_products(of, ob, ref_f, ref_b)
    if M <= 64:
        # the step of the stored passes (round 5): every scan in float - the sums over the states above as native suffix scans -
        # the vector and the diagonal term in fp64.  The off-diagonal part of T carries <= 1e-2 of a row's mass, so its float
        # rounding enters at 1e-2 x 6e-8 of the vector's largest entry; an entry far below that largest entry is only good to that
        # absolute bound (the stored alpha is a float floored at 1e-10 of the sum)
        rc, of, ob = ss_apply(T, x, e, float_scans=True)
        assert rc == 0
        for o, r in ((of, ref_f), (ob, ref_b)):
            assert np.all(np.isfinite(o))
            assert np.max(np.abs(o - r) / np.max(np.abs(r), axis=1, keepdims=True)) < 2e-8


@pytest.mark.parametrize("M", [70, 100, 130, 200])
def test_one_position_other_state_counts(M):
    """State counts that do not fill the lanes (2 .. 4 states per lane, padded): a T of the reference's form from generators."""
    rng = np.random.default_rng(M)
    c0 = 1e-5 / (M + 1)
    g = 1e-4 * rng.random(M)
    u = 1e-3 * rng.random(M)
    r = np.cumsum(0.05 + 0.1 * rng.random(M))
    T = np.zeros((M, M))
    for i in range(M):
        T[i, :i] = g[:i] + c0
        for j in range(i + 1, M):
            T[i, j] = c0 + u[i] * np.exp(-(r[j] - r[i]))
        T[i, i] = 1.0 - T[i].sum()
    x = rng.random((5, M)) ** 6 + 1e-10
    e = 0.1 + rng.random((5, M))
    rc, of, ob = ss_apply(T, x, e)
    assert rc == 0
    ref_f = e * (x @ T)
    ref_b = (e * x) @ T.T
    check_products(of, ob, ref_f, ref_b)


def test_unstructured_matrix_is_refused():
    rng = np.random.default_rng(3)
    M = 24
    T = rng.random((M, M)) + 0.01
    T /= T.sum(axis=1, keepdims=True)
    rc, _, _ = ss_apply(T, np.ones((1, M)), np.ones((1, M)))
    assert rc == 2


def test_engine_falls_back_to_dense_kernels_on_unstructured_T():
    """set_raw with a matrix of no structure: same manager, dense chain kernels, results against the C restatement."""
    from oracle import oracle
    from smcpp_amd import _smcpp
    g = load_golden("G1_M16_n4")
    M = len(g["pi"])
    rng = np.random.default_rng(11)
    S = rng.random((M, M)); S = S + S.T + 20.0 * np.eye(M)      # reversible chain: real spectrum (the oracle takes LAPACK's eig)
    T = S / S.sum(axis=1, keepdims=True)
    obs = np.ascontiguousarray(g["obs"][:800], dtype=np.int32)
    obs[:, 0] = np.minimum(obs[:, 0], 40)
    im = _smcpp.PyOnePopInferenceManager(int(g["n"]), [obs], g["hs"], ("pop1",), float(g["pol"]))
    assert im.chain_mode() == 5
    im.theta = float(g["theta"]); im.rho = float(g["rho"])
    im.set_raw(g["pi"], T, g["keys"], g["E"])
    im.E_step()
    o = oracle.estep(g["pi"], T, g["keys"], g["E"], obs)
    assert abs(im.loglik() - o["loglik"]) <= 1e-6 * abs(o["loglik"])
    assert np.max(np.abs(im.xisums[0] - o["xisum"]) / np.abs(o["xisum"])) <= 5e-6
    # ... and the structured T of the golden on the same manager afterwards (scan kernels)
    im.set_raw(g["pi"], g["T"], g["keys"], g["E"])
    im.E_step()
    o = oracle.estep(g["pi"], g["T"], g["keys"], g["E"], obs)
    assert abs(im.loglik() - o["loglik"]) <= 1e-6 * abs(o["loglik"])
    assert np.max(np.abs(im.xisums[0] - o["xisum"]) / np.abs(o["xisum"])) <= 5e-6


def _cert_run(engine_opt, cert, case):
    """One scenario of the certificate test below under SMCPP_SS_CERT_PASS = cert: -> list of (loglik, xisum, gamma sums) per E-step
    and the plan of the last one."""
    from smcpp_amd import _smcpp, synth
    from smcpp_amd.model import PiecewiseModel
    engine_opt("SMCPP_SS_CERT_PASS", "1" if cert else None)
    g = load_golden("G4_M64_n20_2Mbp")
    a0 = np.array(g["a"], dtype=float)
    out = []
    if case == "default_chunks":
        # 30 Mbp: ~300 000 positions over the default chunk list (one chunk per SIMD and direction), default tolerances
        obs = [synth.synth_contig(0, 30_000_000, 20), synth.synth_contig(5, 3_000_000, 20)]
        steps = [a0, a0, a0 * 1.01]
        warm, chunk, eps = False, 0, (0.0, 0.0)
    elif case == "second_round":
        # 2 Mbp in 150-row chunks with tolerances so tight that a chunk only stops re-running when its input is bitwise what it last
        # ran from: the fixed point degenerates to the sequential algorithm (chunk c exact after c passes) - far more passes than the
        # first round launches, i.e. the branch that launches further rounds and the certificate at the pass limit
        obs = [np.ascontiguousarray(g["obs"], dtype=np.int32)]
        steps = [a0, a0]
        warm, chunk, eps = False, 150, (1e-30, 1e-30)
    else:
        # warm start: passes numbered from 1 or 2, one light pass fewer, unequal light-pass counts of the two directions
        obs = [synth.synth_contig(0, 30_000_000, 20)]
        rng = np.random.default_rng(11)
        steps = [a0] + [a0 * (1.0 + 0.02 * rng.standard_normal(len(a0))) for _ in range(3)] + [a0[::-1] * 2.5]
        warm, chunk, eps = True, 800, (0.0, 0.0)
    im = _smcpp.PyOnePopInferenceManager(20, obs, g["hs"], ("pop1",), float(g["pol"]))
    im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
    if chunk or eps[0]:
        im.set_chunking(chunk, *eps)
    if warm:
        im.set_warm_start(True)
    m = PiecewiseModel(a0, g["s"], 1e4, "pop1")
    im.model = m
    plans = []
    for a in steps:
        m[:] = a
        im.E_step()
        assert im.chain_mode() == 5
        out.append((np.array(im.logliks()), [x.copy() for x in im.xisums], [dict(d) for d in im.gamma_sums]))
        plans.append(im.describe()["plan"])
    return out, plans


@pytest.mark.parametrize("case", ["default_chunks", "second_round", "warm_start"])
def test_certificate_from_flags_equals_the_launched_certificate_pass(engine_opt, case):
    """since round 5 the all-skip pass that certifies convergence is no longer launched - run_chains_ss accepts
    "the last launched pass rewrote no end vector" (engine_plans.hpp).  SMCPP_SS_CERT_PASS=1 launches that pass again.  Both must
    give BITWISE the same log-likelihoods and statistics - a regression would hand unconverged rows to the speculatively queued
    statistics - on the default chunk list, on an input that needs further rounds of launches (and reaches the certificate far
    beyond the first round), and on a warm-started trajectory."""
    flags, plan_f = _cert_run(engine_opt, False, case)
    launched, plan_l = _cert_run(engine_opt, True, case)
    assert not any(p["certificate_pass_launched_up_front"] for p in plan_f[:1]) and all(p["certificate_pass_launched_up_front"] for p in plan_l)
    if case == "second_round":
        # the first round launches 6 passes; the certificate comes much later
        assert plan_f[0]["passes_to_certificate"] > 8 and plan_f[0]["passes_launched"] > 8, plan_f[0]
    for (l0, x0, g0), (l1, x1, g1) in zip(flags, launched):
        assert np.array_equal(l0, l1)
        for a, b in zip(x0, x1):
            assert np.array_equal(a, b)
        for da, db in zip(g0, g1):
            assert sorted(da) == sorted(db)
            for k in da:
                assert np.array_equal(da[k], db[k]), k


def test_mid_size_contig_two_wavefronts_per_simd_and_float_halo(engine_opt):
    """Round 6 (last session): one state per lane, 1.35 - 12 million positions - the chunks are entered through a float halo
    (engine_manager.hpp: make_chunks) instead of light passes, from 1.95 million positions on by two wavefronts per SIMD.  A 150 Mbp
    contig, a 250 Mbp contig (2.5 million positions, 589 000 rows) and a two-contig shard: the new plan against the plan of rounds 3 - 5 (SMCPP_SS_WPC=1, SMCPP_SS_HALO=0) and against
    the SEQUENTIAL algorithm (one chunk per contig and direction: no history, no fixed point) - log-likelihood, xi sums, gamma sums."""
    from smcpp_amd import _smcpp, synth
    from smcpp_amd.model import PiecewiseModel
    g = load_golden("G4_M64_n20_2Mbp")
    a0 = np.array(g["a"], dtype=float)

    def run(obs, chunk=0):
        im = _smcpp.PyOnePopInferenceManager(20, obs, g["hs"], ("pop1",), float(g["pol"]))
        im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
        if chunk:
            im.set_chunking(chunk)
        im.model = PiecewiseModel(a0, g["s"], 1e4, "pop1")
        im.E_step()
        assert im.chain_mode() == 5
        return np.array(im.logliks()), [x.copy() for x in im.xisums], [dict(d) for d in im.gamma_sums], im.describe()["plan"]

    for obs, wpc in (([synth.synth_contig(3, 150_000_000, 20)], 1), ([synth.synth_contig(0, 250_000_000, 20)], 2),
                     ([synth.synth_contig(1, 200_000_000, 20), synth.synth_contig(2, 90_000_000, 20)], 2)):
        ll, xs, gs, plan = run(obs)
        assert plan["wavefronts_per_simd"] == wpc and plan["halo_pass"] and plan["light_passes_forward"] == 0, plan
        seq = run(obs, chunk=10 ** 9)
        assert seq[3]["chunks_forward"] == len(obs), seq[3]
        engine_opt("SMCPP_SS_WPC", "1"); engine_opt("SMCPP_SS_HALO", "0")
        old = run(obs)
        engine_opt("SMCPP_SS_WPC", None); engine_opt("SMCPP_SS_HALO", None)
        assert old[3]["wavefronts_per_simd"] == 1 and not old[3]["halo_pass"], old[3]
        for name, ref in (("sequential", seq), ("rounds 3-5 plan", old)):
            dl = np.max(np.abs(ll - ref[0]) / np.abs(ref[0]))
            dx = max(np.max(np.abs(x - y)) / np.max(np.abs(y)) for x, y in zip(xs, ref[1]))
            dg = max(np.max(np.abs(d[k] - e[k])) / max(np.max(np.abs(e[k])), 1e-300) for d, e in zip(gs, ref[2]) for k in e)
            print(f"{len(obs)} contig(s), {plan['positions']} positions, new plan vs {name}: loglik {dl:.2e}, xi sums {dx:.2e}, gamma sums {dg:.2e}")
            assert dl <= 2e-9 and dx <= 2e-6 and dg <= 2e-6
