"""Soak: thousands of evals on one manager; log-likelihood must not drift and device memory must not grow."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from smcpp_amd import _smcpp, synth
from smcpp_amd.model import PiecewiseModel
M, n = 64, 20
hs = synth.hidden_states(M); a, s = synth.model_pieces()
c = synth.synth_contig(0, 100_000_000, n)
_smcpp.set_num_threads(12)
im = _smcpp.PyOnePopInferenceManager(n, [c], hs, ("pop1",), 0.5)
im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
rng = np.random.RandomState(0)
free0 = None
lls = {}
t0 = time.time()
N = int(os.environ.get("SOAK_EVALS", 4000))
for it in range(N):
    k = it % 4                                   # four models in rotation: parameters really change between evals
    m = PiecewiseModel(a * (1.0 + 0.05 * k), s, 1e4, "pop1")
    im.model = m; im.E_step(); ll = im.loglik()
    if k in lls:
        assert ll == lls[k], (it, k, ll, lls[k])  # bit-identical for identical parameters
    else:
        lls[k] = ll
    if it == 50:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print(f"{N} evals in {time.time() - t0:.1f} s, logliks {sorted(lls.values())}, device memory delta {(free0 - free1) / 1e6:.1f} MB")
assert abs(free0 - free1) < 64e6
print("soak ok")
