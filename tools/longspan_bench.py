import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smcpp_amd import _smcpp, synth
from smcpp_amd import _engine as E   # (the engine parses SMCPP_* once per process: switches go through E.set_option)
from smcpp_amd.model import PiecewiseModel
M, n = 64, 20
hs = synth.hidden_states(M); a, s = synth.model_pieces()
c = synth.synth_contig(0, 100_000_000, n).copy()
rng = np.random.RandomState(1)
lr = np.nonzero(c[:, 0] > 1)[0]
pick = rng.choice(lr, size=len(lr) // 25, replace=False)
c[pick, 0] = rng.randint(32, 3000, size=len(pick))
_smcpp.set_num_threads(8)
for mode in ("0", "1"):
    E.set_option("SMCPP_POWER_PREPASS", mode)
    im = _smcpp.PyOnePopInferenceManager(n, [c], hs, ("pop1",), 0.5)
    im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
    m = PiecewiseModel(a, s, 1e4, "pop1")
    for _ in range(3):
        im.model = m; im.E_step(); im.loglik()
    t = time.perf_counter()
    for _ in range(20):
        im.model = m; im.E_step(); ll = im.loglik()
    dt = (time.perf_counter() - t) / 20
    print("prepass", mode, "ms/eval %.3f" % (dt * 1e3), "loglik", ll, "timing", im.last_timing())
