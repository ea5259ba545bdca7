"""Which device allocation does a kernel read before anything wrote it?  (run through gpurun)

    python tools/poison_probe.py [case ...]

hipMalloc hands out zeroed pages in a fresh process and recycled ones later, so such a read is invisible to a test that runs alone
and shows up as an order-dependent failure in a long-lived process.  For every case (a golden E-step with / without save_gamma, a
model-parameter E-step with gradients, two populations, M = 256) the probe first logs the allocations of one manager
(SMCPP_DEBUG_POISON_LOG), then builds one manager per allocation index with ONLY that allocation filled with 0xFF bytes
(NaN / -1) and compares loglik, xi sums, gamma sums, gamma and Q with the clean run.  Prints the source lines of the offenders."""
import io, os, re, subprocess, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from smcpp_amd import _engine as E, _smcpp, synth
from smcpp_amd.model import PiecewiseModel

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def golden(name):
    z = np.load(os.path.join(G, name + ".npz"))
    return {k: z[k] for k in z.files}


def run_case(case):
    """-> dict of outputs of one fresh manager"""
    kind, name = case.split(":")
    g = golden(name) if not name.startswith("params") else dict(np.load(os.path.join(G, name + ".npz")))
    if kind == "cut":
        # (round 6) rows longer than 64 positions cut into pieces, M = 144 (nine 16-state tiles: the LDS-staged rank updates with a ragged
        # block), save_gamma: per-row posteriors from scan steps
        obs = [np.ascontiguousarray(g["obs"][:1500], dtype=np.int32)]
        a_, s_ = synth.model_pieces()
        im = _smcpp.PyOnePopInferenceManager(int(g["n"]), obs, synth.hidden_states(144), ("pop1",), 0.5)
        im.model = PiecewiseModel(a_, s_, 1e4, "pop1")
        im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = 1.0
        im.save_gamma = True
    elif kind == "m768":
        # sixteen states per lane, ONE chunk per contig (the sequential algorithm)
        n = 10
        obs = [np.ascontiguousarray(synth.synth_contig(0, 100_000_000, n)[:70], dtype=np.int32)]
        a_, s_ = synth.model_pieces()
        im = _smcpp.PyOnePopInferenceManager(n, obs, synth.hidden_states(768), ("pop1",), 0.5)
        im.model = PiecewiseModel(a_, s_, 1e4, "pop1")
        im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
        im.set_chunking(10 ** 6)
    elif kind == "post128":
        # (round 6) un-binned rows at M = 128: dense streamed chains, eigensystem statistics, per-row posteriors from eigen-power pieces
        obs = [np.ascontiguousarray(synth.synth_posterior_contig(3000, 8, seed=7), dtype=np.int32)]
        a_, s_ = synth.model_pieces()
        im = _smcpp.PyOnePopInferenceManager(8, obs, synth.hidden_states(128), ("pop1",), 0.5)
        im.model = PiecewiseModel(a_, s_, 1e4, "pop1")
        im.theta = 2e-4; im.rho = 6e-5; im.alpha = 1.0
        im.save_gamma = True
    elif kind in ("big", "biggamma", "post", "m1"):
        # big: a whole 100 Mbp contig (512 chunks per direction, light passes); post: un-binned rows (hybrid chains, thousands of
        # span groups); m1: ONE hidden state (the bootstrap manager of Analysis)
        n = int(g["n"])
        if kind == "post":
            obs = [np.ascontiguousarray(synth.synth_posterior_contig(200_000, 8, seed=7), dtype=np.int32)]
            hs, n = synth.hidden_states(32), 8
        elif kind == "m1":
            obs = [synth.synth_contig(3, 2_000_000, n)]
            hs = np.array([0.0, np.inf])
        else:
            obs = [synth.synth_contig(0, 100_000_000, n)]
            hs = g["hs"]
        im = _smcpp.PyOnePopInferenceManager(n, obs, hs, ("pop1",), float(g["pol"]))
        im.theta = float(g["theta"]) if kind != "post" else 2e-4
        im.rho = float(g["rho"]) if kind != "post" else 6e-5
        im.alpha = 1.0
        im.model = PiecewiseModel(g["a"], g["s"], 1e4, "pop1")
        im.save_gamma = kind in ("biggamma", "post")
    elif kind in ("raw", "gamma"):
        obs = np.ascontiguousarray(g["obs"], dtype=np.int32)
        if obs.shape[1] == 4:
            im = _smcpp.PyOnePopInferenceManager(int(g["n"]), [obs], g["hs"], ("pop1",), float(g["pol"]))
        else:
            im = _smcpp.PyTwoPopInferenceManager(10, 10, 2, 0, [obs], g["hs"], ("pop1", "pop2"), float(g["pol"]))
        im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
        im.set_raw(g["pi"], g["T"], g["keys"], g["E"])
        im.save_gamma = kind == "gamma"
    else:                                       # model: the engine's own cold preparation, two contigs, Q with gradient
        n = int(g["n"])
        obs = [synth.synth_contig(1, 3_000_000, n), synth.synth_contig(2, 400_000, n)]
        im = _smcpp.PyOnePopInferenceManager(n, obs, g["hs"], ("pop1",), float(g["pol"]))
        im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
        im.model = PiecewiseModel(g["a"], g["s"], 1e4, "pop1")
    out = {}
    for rep in range(2):                        # two E-steps: the second one takes the adapted launch count
        im.E_step()
        out[f"ll{rep}"] = np.array(im.logliks())
    out["xisum"] = np.array(im.xisums)
    gs = im.gamma_sums
    out["gsum"] = np.array([gs[c][k] for c in range(len(gs)) for k in sorted(gs[c])])
    out["gamma"] = np.concatenate([x.ravel() for x in im.gammas])
    out["q"] = np.array(im.Q(separate=True))
    if kind == "model":
        q, jac = im.Q_with_gradient() if hasattr(im, "Q_with_gradient") else (None, None)
        if jac is not None:
            out["jac"] = np.asarray(jac)
    if kind in ("gamma", "biggamma", "post", "cut"):
        out["argmax"] = np.asarray(im.gamma_argmax(0)).astype(np.float64)
    return out


def main():
    cases = sys.argv[1:] or ["raw:G1_M16_n4", "raw:G3_M32_n10_2Mbp", "raw:G4_M64_n20_2Mbp", "gamma:G3_M32_n10_2Mbp", "gamma:G7_M32_n8_chr11",
                             "gamma:G18_M64_n8_chr11", "raw:G5_M48_twopop_layout", "raw:G2_M51_n6_longspans", "model:params_M64_n20",
                             "model:params_M32_n10", "model:params_M256_n50", "m1:params_M32_n10", "big:params_M64_n20",
                             "biggamma:params_M32_n10", "post:params_M32_n10", "big:params_M256_n50", "cut:G1_M16_n4", "biggamma:params_M256_n50"]
    for case in cases:
        if os.environ.get("PROBE_CHILD") == case:
            # child: log the allocations of one clean manager to stderr
            E.set_option("SMCPP_DEBUG_POISON_LOG", "1")
            run_case(case)
            return
    for case in cases:
        env = dict(os.environ, PROBE_CHILD=case)
        log = subprocess.run([sys.executable, os.path.abspath(__file__), case], env=env, capture_output=True, text=True).stderr
        allocs = [(int(m.group(1)), m.group(2), int(m.group(3))) for m in re.finditer(r"\[alloc (\d+)\] (\S+) (\d+) bytes", log)]
        E.set_option("SMCPP_DEBUG_POISON", None)
        clean = run_case(case)
        bad = []
        for idx, where, nbytes in allocs:
            E.set_option("SMCPP_DEBUG_POISON", "255")
            E.set_option("SMCPP_DEBUG_POISON_ONLY", str(idx))
            try:
                got = run_case(case)
                diff = [k for k in clean if got[k].shape != clean[k].shape or not np.array_equal(got[k], clean[k], equal_nan=False)]
            except Exception as ex:                           # noqa: BLE001
                diff = ["EXCEPTION " + str(ex)[:80]]
            if diff:
                bad.append((idx, where, nbytes, diff))
        E.set_option("SMCPP_DEBUG_POISON", None); E.set_option("SMCPP_DEBUG_POISON_ONLY", None)
        print(f"== {case}: {len(allocs)} allocations, {len(bad)} read before written", flush=True)
        for idx, where, nbytes, diff in bad:
            print(f"   alloc {idx:3d} {os.path.basename(where)} {nbytes} bytes -> differs: {diff}", flush=True)


if __name__ == "__main__":
    main()
