"""Many small contigs on one GPU (M = 64, n = 20): the chunk plan the engine picks against the plan of rounds 3-5 (SMCPP_SS_WPC=1,
SMCPP_SS_HALO=0).   python tools/multi_contig_probe.py   (GPU box)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smcpp_amd import _smcpp, synth
from smcpp_amd import _engine as E
from smcpp_amd.model import PiecewiseModel
M, n = 64, 20
hs = synth.hidden_states(M); a, s = synth.model_pieces()
_smcpp.set_num_threads(12)
m = PiecewiseModel(a, s, 1e4, "pop1")
for name, lens in (("10 x 20 Mbp", [20] * 10), ("22 x 10 Mbp", [10] * 22), ("6 x 60 Mbp", [60] * 6), ("40 x 15 Mbp", [15] * 40), ("3 x 150 Mbp", [150] * 3),
                   ("300 x 1 Mbp", [1] * 300), ("1500 x 0.2 Mbp", [0.2] * 1500), ("1 x 120 + 60 x 1 Mbp", [120] + [1] * 60)):
    contigs = [synth.synth_contig(i, int(L * 1e6), n) for i, L in enumerate(lens)]
    res = {}
    for mode in ("new", "old"):
        E.set_option("SMCPP_SS_WPC", "1" if mode == "old" else None)
        E.set_option("SMCPP_SS_HALO", "0" if mode == "old" else None)
        im = _smcpp.PyOnePopInferenceManager(n, contigs, hs, ("pop1",), 0.5)
        im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
        for _ in range(3):
            im.model = m; im.E_step(); im.loglik()
        t = time.perf_counter()
        for _ in range(16):
            im.model = m; im.E_step(); ll = im.loglik()
        ms = (time.perf_counter() - t) / 16 * 1e3
        p = im.describe()["plan"]
        res[mode] = (ms, ll)
        print(f"{name} ({p['positions']} positions) {mode}: {ms:.3f} ms per eval, wavefronts per SIMD {p['wavefronts_per_simd']}, halo {p['halo_pass']}, "
              f"passes {p['passes_launched']}, light {p['light_passes_forward']}/{p['light_passes_backward']}, loglik {ll:.6f}", flush=True)
        del im
    print(f"   new / old = {res['new'][0] / res['old'][0]:.3f}, loglik rel diff {abs(res['new'][1] - res['old'][1]) / abs(res['old'][1]):.2e}", flush=True)
E.set_option("SMCPP_SS_WPC", None); E.set_option("SMCPP_SS_HALO", None)
