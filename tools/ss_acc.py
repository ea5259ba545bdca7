"""Accuracy probe of the scan chains (GPU box): statistics against the C restatement for several light-pass settings."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from smcpp_amd import _smcpp, synth
from smcpp_amd import _engine as E   # (the engine parses SMCPP_* once per process: switches go through E.set_option)
from smcpp_amd.model import PiecewiseModel

M, n = int(sys.argv[1]), int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000_000
hs = synth.hidden_states(M)
a, s = synth.model_pieces()
contigs = [synth.synth_contig(300 + M, L, n), synth.synth_contig(301 + M, 150_000, n)]
ref = None
for lf, lb, eps in [(0, 0, None), (2, 2, None), (2, 2, (2e-7, 1e-7)), (3, 3, None), (1, 1, None)]:
    E.set_option("SMCPP_SS_LIGHT_F", lf); E.set_option("SMCPP_SS_LIGHT_B", lb)
    im = _smcpp.PyOnePopInferenceManager(n, contigs, hs, ("pop1",), 0.5)
    im.model = PiecewiseModel(a, s, 1e4, "pop1")
    im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
    if eps: im.set_chunking(0, eps[0], eps[1])
    im.E_step()
    if ref is None:
        pi, T, keys = im.pi, im.transition, im.keys
        ep = im.emission_probs
        Etab = np.array([ep[tuple(k)] for k in keys.tolist()])
        ref = [oracle.estep(pi, T, keys, Etab, c) for c in contigs]
    t = im.last_timing()
    for c in range(len(contigs)):
        xs = im.xisums[c]
        re = np.abs(xs - ref[c]["xisum"]) / np.abs(ref[c]["xisum"])
        i, j = np.unravel_index(np.argmax(re), re.shape)
        print(f"light {lf}/{lb} eps {eps} contig {c}: passes {t['fwd_passes']}, loglik rel {abs(im.logliks()[c]-ref[c]['loglik'])/abs(ref[c]['loglik']):.2e}, "
              f"xisum max rel {re.max():.2e} at ({i},{j}) val {ref[c]['xisum'][i,j]:.3e}; median {np.median(re):.2e}; [0,0] {re[0,0]:.2e}")
