"""Developer tool (not collected by pytest): accuracy / time of the chunk-boundary tolerances on the headline workload.
The single-chunk run (= the purely sequential algorithm) is the yardstick.  Usage: python tools/gpu_eps_sweep.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smcpp_amd import _smcpp, synth

par = np.load(os.path.join(ROOT, "tests", "golden", "params_M64_n20.npz"))
obs = synth.synth_contig(0, 100_000_000, 20)
im = _smcpp.PyOnePopInferenceManager(20, [obs], par["hs"], ("pop1",), float(par["pol"]))
im.theta = float(par["theta"]); im.rho = float(par["rho"]); im.alpha = float(par["alpha"])


def run(chunk, ea, eb, reps=5):
    im.set_chunking(chunk, ea, eb)
    ts = []
    for _ in range(reps):
        im.set_raw(par["pi"], par["T"], par["keys"], par["E"])
        t0 = time.perf_counter(); im.E_step(); ll = im.loglik(); ts.append(time.perf_counter() - t0)
    t = im.last_timing()
    xs = im.xisums[0].copy()
    gs = im.gamma_sums[0]
    return ll, xs, gs, min(ts) * 1e3, t


ll0, xs0, gs0, _, _ = run(10 ** 9, 0, 0, reps=1)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


for ea, eb in [(2e-6, 1e-8), (2e-6, 1e-6), (2e-6, 2e-6), (2e-6, 4e-6), (4e-6, 4e-6), (2e-6, 1e-5), (1e-5, 1e-5)]:
    ll, xs, gs, ms, t = run(0, ea, eb)
    ge = max(rel(gs[k], gs0[k]) for k in gs0)
    print(f"eps_a={ea:g} eps_b={eb:g}: {ms:.2f} ms  fwd {t['forward_ms']:.2f} ({t['fwd_passes']:.0f}) bwd {t['backward_ms']:.2f} "
          f"({t['bwd_passes']:.0f})  dll_rel={abs(ll - ll0) / abs(ll0):.2e} xisum_rel={rel(xs, xs0):.2e} gsum_rel={ge:.2e}", flush=True)
