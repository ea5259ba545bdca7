"""Developer probe (not collected): error of the chunked chains on the binned example contig against the C restatement."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle
from smcpp_amd import _smcpp, data as D, synth, vcf2smc as V
from smcpp_amd.model import PiecewiseModel
c, _ = V.vcf2smc(os.path.join(ROOT, "tests", "golden", "example.vcf.gz"), "1", ("pop1", ["msp_0", "msp_1", "msp_2"]))
piece = D.break_long_spans(D.Contig(D.compress_repeated_obs(c.data), c.pid, c.n, c.a), 100000)[0]
binned = D.recode_monomorphic(D.Contig(D.bin_observations(D.thin_data(piece.data, 895), 100, [2]), c.pid, c.n, c.a))
thin = np.ascontiguousarray(D.compress_repeated_obs(binned.data), dtype=np.int32)
a, s = synth.model_pieces()
hs = synth.hidden_states(15)
for theta, rho in [(0.1, 0.025), (0.0402, 0.01)]:
    for chunk, ea, eb in [(10 ** 9, 0, 0), (0, 0, 0), (0, 2e-6, 1e-7), (0, 5e-7, 1e-6), (0, 5e-7, 1e-8), (256, 0, 0)]:
        im = _smcpp.PyOnePopInferenceManager(4, [thin], hs, ("pop1",), 0.5)
        im.model = PiecewiseModel(a, s, 1e4, "pop1")
        im.theta = theta; im.rho = rho; im.alpha = 1.0
        im.set_chunking(chunk, ea, eb)
        im.E_step()
        ep = im.emission_probs
        Etab = np.array([ep[tuple(k)] for k in im.keys.tolist()])
        o = oracle.estep(im.pi, im.transition, im.keys, Etab, thin)
        x = im.xisums[0]
        rel = np.abs(x - o["xisum"]) / np.abs(o["xisum"])
        t = im.last_timing()
        print(f"theta={theta} chunk={chunk} eps=({ea:g},{eb:g}): xisum max rel {rel.max():.2e} at {np.unravel_index(rel.argmax(), rel.shape)} "
              f"ll rel {abs(im.loglik()-o['loglik'])/abs(o['loglik']):.1e} passes {t['fwd_passes']:.0f}/{t['bwd_passes']:.0f}", flush=True)
