"""Golden G14: config C5 (M = 256, n = 50) on the first 5 000 rows of its 100 Mbp contig, from the COMPILED REFERENCE
(oracle/_ref: the reference's own src/hmm.cpp ... built where they lie by oracle/Makefile; build container only):

    make -C oracle ref && python tests/golden/make_golden_c5.py

The parameters (pi, T, emission table) are those of tests/golden/params_M256_n50.npz (reference `ref_prep` + the
emission assembly of oracle/prep_oracle.py); this file adds the E-step's outputs on a slice long enough for the
chunk-parallel chains to iterate (the C restatement needs ~0.1 s per eigen row at this M, the reference 7 ms).
Data only: rows (int32) and the reference's loglik / Q / xisum / gamma sums / gamma[:,0].
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
    full = synth.synth_contig(0, 100_000_000, 50)
    obs = np.ascontiguousarray(full[:5000], dtype=np.int32)
    known = {tuple(int(x) for x in k) for k in g["keys"]}
    assert all(tuple(r) in known for r in obs[:, 1:].tolist())
    t = time.time()
    r = ref.estep(g["pi"], g["T"], g["keys"], g["E"], obs)
    print("reference E-step: %.1f s" % (time.time() - t))
    gs_keys = np.array(list(r["gamma_sums"].keys()), dtype=np.int32)
    gs_vals = np.array(list(r["gamma_sums"].values()))
    out = dict(obs=obs, loglik=np.array(r["loglik"]), q=r["q"], xisum=r["xisum"], gs_keys=gs_keys, gs_vals=gs_vals,
               gamma0=r["gamma"][:, 0].copy())
    path = os.path.join(HERE, "G14_c5_slice.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; loglik", float(r["loglik"]), "rows", len(obs), "positions", int(obs[:, 0].sum()))


if __name__ == "__main__":
    main()
