"""What one rank of the 8-GPU whole-genome run (configs[2]) does, on one GPU: its LPT shard of the 22 contigs."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smcpp_amd import _smcpp, synth, dist as sd
from smcpp_amd import _engine as E   # (the engine parses SMCPP_* once per process: switches go through E.set_option)
from smcpp_amd.model import PiecewiseModel
M, n = 64, 20
hs = synth.hidden_states(M); a, s = synth.model_pieces()
world = int(os.environ.get("SHARD_WORLD", 8))
owner = sd.lpt_shard(synth.C3_LENGTHS_MBP, world)
_smcpp.set_num_threads(12)
m = PiecewiseModel(a, s, 1e4, "pop1")
for rank in range(min(world, int(os.environ.get("SHARD_RANKS", 2)))):
    idx = [i for i in range(len(owner)) if owner[i] == rank]
    contigs = [synth.synth_contig(i, int(synth.C3_LENGTHS_MBP[i] * 1e6), n) for i in idx]
    rows = sum(len(c) for c in contigs)
    for mode in os.environ.get("SHARD_MODES", "ss,coop,lock").split(","):
        E.set_option("SMCPP_CHAIN", None)
        if mode != "ss":                      # "ss" = the engine's own choice (scan chains)
            E.set_option("SMCPP_CHAIN", mode)
        im = _smcpp.PyOnePopInferenceManager(n, contigs, hs, ("pop1",), 0.5)
        im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
        for _ in range(3):
            im.model = m; im.E_step(); im.loglik()
        t = time.perf_counter()
        for _ in range(20):
            im.model = m; im.E_step(); ll = im.loglik()
        ms = (time.perf_counter() - t) / 20 * 1e3
        tm = im.last_timing()
        print(f"world {world} rank {rank}: contigs {idx} rows {rows} {mode}: {ms:.2f} ms per eval", {k: round(float(v), 2) for k, v in tm.items()})
        del im
