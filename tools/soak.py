"""Soak: thousands of evals on one manager; the log-likelihood (and, with save_gamma, the decoded index of every column) must be
bit-identical whenever the parameters are, and device memory must not grow.   python tools/soak.py   (GPU box; a minute)
Configurations (round 6): the headline contig at M = 64 with and without save_gamma (eigen-free per-row posteriors), config C5's contig
at M = 256 (LDS-staged rank updates, lazily expanded transition matrix) with and without save_gamma, and the example-derived contig at
M = 144 (rows cut into pieces, a ragged output block), un-binned rows at M = 128 (per-row posteriors from eigen-power pieces)."""
import os, sys, time, zlib, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from smcpp_amd import _smcpp, synth
from smcpp_amd.model import PiecewiseModel

_smcpp.set_num_threads(12)
a, s = synth.model_pieces()
SCALE = float(os.environ.get("SOAK_SCALE", 1.0))


def soak(name, M, n, contig, evals, save_gamma, theta=synth.THETA, rho=synth.RHO):
    im = _smcpp.PyOnePopInferenceManager(n, [contig], synth.hidden_states(M), ("pop1",), 0.5)
    im.theta = theta; im.rho = rho; im.alpha = 1.0
    im.save_gamma = save_gamma
    seen, free0 = {}, None
    t0 = time.time()
    N = max(8, int(evals * SCALE))
    for it in range(N):
        k = it % 4                                   # four models in rotation: parameters really change between evals
        im.model = PiecewiseModel(a * (1.0 + 0.05 * k), s, 1e4, "pop1")
        im.E_step()
        got = (im.loglik(),) + ((zlib.crc32(np.asarray(im.gamma_argmax(0)).tobytes()),) if save_gamma and it % 16 < 4 else ())
        key = (k, len(got))
        if key in seen:
            assert got == seen[key], (name, it, k, got, seen[key])      # bit-identical for identical parameters
        else:
            seen[key] = got
        if it == min(50, N // 2):
            free0 = torch.cuda.mem_get_info()[0]
    free1 = torch.cuda.mem_get_info()[0]
    print(f"{name}: {N} evals in {time.time() - t0:.1f} s, logliks {sorted(v[0] for kk, v in seen.items() if kk[1] == 1 or not save_gamma)[:4]}, "
          f"device memory delta {(free0 - free1) / 1e6:.1f} MB, plan {im.describe()['plan']['per_row_gamma']}", flush=True)
    assert abs(free0 - free1) < 64e6


g1 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G1_M16_n4.npz"))
c64 = synth.synth_contig(0, 100_000_000, 20)
c256 = synth.synth_contig(0, 100_000_000, 50)
soak("headline M=64", 64, 20, c64, 4000, False)
soak("headline M=64 save_gamma", 64, 20, c64, 1500, True)
soak("c5 M=256", 256, 50, c256, 1200, False)
soak("c5 M=256 save_gamma", 256, 50, c256, 400, True)
soak("example-derived M=144 cut rows save_gamma", 144, 4, np.ascontiguousarray(g1["obs"], dtype=np.int32), 1500, True,
     theta=float(g1["theta"]), rho=float(g1["rho"]))
# (round 6, last session) un-binned rows at M = 128: per-row posteriors from eigen-power pieces + scan steps
soak("un-binned M=128 save_gamma (eigen-power pieces)", 128, 8, np.ascontiguousarray(synth.synth_posterior_contig(20000, 8, seed=7), dtype=np.int32),
     300, True, theta=2e-4, rho=6e-5)
# (round 6, last session) a mid-size contig: two wavefronts per SIMD through a float halo (engine_manager.hpp: make_chunks)
soak("250 Mbp M=64 (mid-size chunk plan)", 64, 20, synth.synth_contig(0, 250_000_000, 20), 600, False)
soak("250 Mbp M=64 save_gamma (mid-size chunk plan)", 64, 20, synth.synth_contig(0, 250_000_000, 20), 200, True)
print("soak ok")
