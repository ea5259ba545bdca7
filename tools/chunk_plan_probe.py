"""The chunk plan the engine picks (`auto`) against the plan of rounds 3-5 (`old`) and forced alternatives, single contigs of 150 / 250 / 700 Mbp at
M = 32 / 128 / 192 (n = 20).   python tools/chunk_plan_probe.py   (GPU box; profiles/r06_g_chunk_plan_probe.log)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smcpp_amd import _smcpp, synth
from smcpp_amd import _engine as E
from smcpp_amd.model import PiecewiseModel
n = 20
a, s = synth.model_pieces()
_smcpp.set_num_threads(12)
m = PiecewiseModel(a, s, 1e4, "pop1")
for M in (32, 128, 192):
    hs = synth.hidden_states(M)
    for L in (150, 250, 700):
        contigs = [synth.synth_contig(0, int(L * 1e6), n)]
        modes = [("auto", {}), ("old", {"SMCPP_SS_WPC": "1", "SMCPP_SS_HALO": "0"} if M <= 64 else {"SMCPP_SS_WPC": "1"}),
                 ("wpc2", {"SMCPP_SS_WPC": "2"}), ("wpc2 float halo", {"SMCPP_SS_WPC": "2", "SMCPP_SS_HALO": "1", "SMCPP_HALO_DF": "0", "SMCPP_HALO_DB": "0"})]
        for name, env in modes:
            for k in ("SMCPP_SS_WPC", "SMCPP_SS_HALO", "SMCPP_HALO_DF", "SMCPP_HALO_DB"):
                E.set_option(k, env.get(k))
            im = _smcpp.PyOnePopInferenceManager(n, contigs, hs, ("pop1",), 0.5)
            im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
            for _ in range(3):
                im.model = m; im.E_step(); im.loglik()
            t = time.perf_counter()
            for _ in range(12):
                im.model = m; im.E_step(); ll = im.loglik()
            ms = (time.perf_counter() - t) / 12 * 1e3
            p = im.describe()["plan"]
            print(f"M={M} L={L} {name}: {ms:.3f} ms, wpc {p['wavefronts_per_simd']}, halo {p['halo_pass']}, passes {p['passes_launched']}, light {p['light_passes_forward']}/{p['light_passes_backward']}, chains {im.last_timing()['chains_wall_ms']:.3f}", flush=True)
            del im
