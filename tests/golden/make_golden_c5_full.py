"""Golden G22: config C5 (M = 256, n = 50) on its WHOLE 100 Mbp contig (235 k rows) from the COMPILED REFERENCE
(oracle/_ref: the reference's own src/hmm.cpp ... built where they lie by oracle/Makefile; build container only):

    make -C oracle ref && python tests/golden/make_golden_c5_full.py

One `HMM::Estep` (src/hmm.cpp:45-153) on one core, as the reference runs a contig (src/inference_manager.cpp:89-94): about
40 minutes here.  G14 / G16_c5 pin the first 5 000 / 25 000 rows; this file pins the last 90 % of the contig, where the
chunk-parallel fixed point of the scan chains runs over a thousand chunks ("What's weak" 3).
Data only: the reference's loglik, Q, xisum, gamma sums and gamma[:, 0]; the rows are NOT stored - the generator is
deterministic and `crc` pins them.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402
from smcpp_amd import synth  # noqa: E402


def main():
    g = dict(np.load(os.path.join(HERE, "params_M256_n50.npz")))
    obs = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, 50), dtype=np.int32)
    known = {tuple(int(x) for x in k) for k in g["keys"]}
    present = {tuple(r) for r in np.unique(obs[:, 1:], axis=0).tolist()}
    assert present <= known
    t = time.time()
    r = ref.estep(g["pi"], g["T"], g["keys"], g["E"], obs)
    dt = time.time() - t
    print("reference E-step: %.1f s" % dt, flush=True)
    keys = [tuple(int(x) for x in k) for k in g["keys"]]
    gs = np.zeros((len(keys), len(g["pi"])))
    have = np.zeros(len(keys), dtype=bool)
    for i, k in enumerate(keys):
        if k in r["gamma_sums"]:
            gs[i] = r["gamma_sums"][k]; have[i] = True
    out = dict(keys=g["keys"], rows=np.array(len(obs)), positions=np.array(int(obs[:, 0].sum())),
               crc=np.array(synth.contig_crc(obs), dtype=np.int64), loglik=np.array(r["loglik"]), q=r["q"],
               xisum=r["xisum"], gs=gs, gs_have=have, gamma0=r["gamma"][:, 0].copy(), ref_seconds=np.array(dt))
    path = os.path.join(HERE, "G22_c5_full.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; loglik", repr(float(r["loglik"])), "rows", len(obs), flush=True)


if __name__ == "__main__":
    main()
