"""Many small contigs (scaffold-level assemblies): where an E-step's time goes as the contig count grows.  python tools/many_contigs_probe.py [nc L_mbp ...]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smcpp_amd import _smcpp, synth
from smcpp_amd.model import PiecewiseModel
M, n = 64, 20
hs = synth.hidden_states(M); a, s = synth.model_pieces()
_smcpp.set_num_threads(12)
m = PiecewiseModel(a, s, 1e4, "pop1")
arg = [float(x) for x in sys.argv[1:]] or [1500, 0.2, 400, 0.75, 100, 3.0, 22, 13.6]
for nc, L in zip(arg[0::2], arg[1::2]):
    nc = int(nc)
    contigs = [synth.synth_contig(i, int(L * 1e6), n) for i in range(nc)]
    im = _smcpp.PyOnePopInferenceManager(n, contigs, hs, ("pop1",), 0.5)
    im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
    for _ in range(3):
        im.model = m; im.E_step(); im.loglik()
    t = time.perf_counter()
    for _ in range(10):
        im.model = m; im.E_step(); ll = im.loglik()
    ms = (time.perf_counter() - t) / 10 * 1e3
    p = im.describe()["plan"]
    print(nc, "x", L, "Mbp:", round(ms, 3), "ms", {k: round(float(v), 3) for k, v in im.last_timing().items()}, p["chunks_forward"], p["chunks_backward"], p["wavefronts_per_simd"], flush=True)
