"""Where do the lock-step chains overtake the cooperative ones?  Whole-genome-like inputs of growing size, both kernels."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smcpp_amd import _smcpp, synth
from smcpp_amd import _engine as E   # (the engine parses SMCPP_* once per process: switches go through E.set_option)
from smcpp_amd.model import PiecewiseModel
M, n = int(os.environ.get("LOCK_M", 64)), int(os.environ.get("LOCK_N", 20))
hs = synth.hidden_states(M); a, s = synth.model_pieces()
L = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51]
allc = [synth.synth_contig(i, l * 1_000_000, n) for i, l in enumerate(L)]
_smcpp.set_num_threads(8)
m = PiecewiseModel(a, s, 1e4, "pop1")
for k in (3, 6, 10, 15, 22):
    contigs = allc[:k]
    rows = sum(len(c) for c in contigs)
    out = []
    for mode in ("coop", "lock"):
        E.set_option("SMCPP_CHAIN", mode)
        im = _smcpp.PyOnePopInferenceManager(n, contigs, hs, ("pop1",), 0.5)
        im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
        for _ in range(2):
            im.model = m; im.E_step(); im.loglik()
        t = time.perf_counter()
        for _ in range(5):
            im.model = m; im.E_step(); ll = im.loglik()
        out.append((time.perf_counter() - t) / 5 * 1e3)
        del im
    print(f"contigs {k:2d} rows {rows:8d} rows/(CU x 16) {rows // 4096:5d}: coop {out[0]:7.2f} ms  lock {out[1]:7.2f} ms")
