import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0)
from smcpp_amd import _engine as E, _smcpp, synth
from smcpp_amd.model import PiecewiseModel
par = np.load("tests/golden/params_M64_n20.npz")
obs = [synth.synth_contig(0, 100_000_000, 20)]
_smcpp.set_num_threads(15)
im = _smcpp.PyOnePopInferenceManager(20, obs, par["hs"], ("pop1",), float(par["pol"]), device=0)
im.theta = float(par["theta"]); im.rho = float(par["rho"]); im.alpha = float(par["alpha"])
a, s = np.ascontiguousarray(par["a"], dtype=np.float64), np.ascontiguousarray(par["s"], dtype=np.float64)
if os.environ.get("FIRST_EVAL"):
    im.model = PiecewiseModel(a, s, 1e4, "pop1"); im.E_step(); im.loglik()
    kk = im.keys; ep = im.emission_probs
    raw = (im.pi, im.transition, kk, np.array([ep[tuple(int(x) for x in k)] for k in kk]))
im.model = PiecewiseModel(a, s, 1e4, "pop1"); im.E_step()
K = len(a); da = np.ascontiguousarray(np.eye(K)); val = np.zeros(4); jac = np.zeros((4, K))
rng = np.random.default_rng(5)
ts, tq = [], []
for i in range(60):
    ai = np.ascontiguousarray(a * np.exp(0.01 * rng.standard_normal(K)))
    t0 = time.perf_counter()
    E.check(E.lib().smcpp_set_params(im._im, K, E.dptr(ai), E.dptr(da), K, E.dptr(s)))
    t1 = time.perf_counter()
    E.check(E.lib().smcpp_q(im._im, E.dptr(val), E.dptr(jac)))
    t2 = time.perf_counter()
    ts.append(t1 - t0); tq.append(t2 - t1)
print(os.environ.get("WITH_TORCH"), os.environ.get("FIRST_EVAL"), "set_params median %.1f us, q median %.1f us" % (1e6 * np.median(ts[10:]), 1e6 * np.median(tq[10:])))
