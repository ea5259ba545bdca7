"""Golden G21: M = 512 hidden states (the round-5 lift of the M <= 256 limit) from the COMPILED REFERENCE (oracle/_ref; build
container only):

    make -C oracle ref && python tests/golden/make_golden_m512.py        -> tests/golden/G21_M512_n10_1000rows.npz
    python tests/golden/make_golden_m512.py 768 300                      -> tests/golden/G24_M768_n10_300rows.npz  (round 6)

`HMM::Estep` (src/hmm.cpp:45-153) on the first 1 000 rows of the synthetic contig 0 (100 bp bins, n = 10) with M = 512 hidden states on
the parameters of `ref_prep` + the emission assembly of oracle/prep_oracle.py - the reference itself has no limit on M
(src/inference_manager.cpp:21-54).  Data only, < 100 KB: the MODEL (a, s, hidden states, theta, rho - the engine prepares pi / T / E
itself), the keys, the reference's loglik, Q, the row / column sums, diagonal and total of xisum (the 512 x 512 matrix itself would be
2 MB), the gamma sums and gamma[:, 0].
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden import params_for  # noqa: E402
from oracle import ref  # noqa: E402
from smcpp_amd import synth  # noqa: E402

M, N, ROWS = 512, 10, 1000
if len(sys.argv) >= 3:          # python make_golden_m512.py 768 300  -> G24_M768_n10_300rows.npz (round 6: 512 < M <= 1024)
    M, ROWS = int(sys.argv[1]), int(sys.argv[2])
TAG = "G21" if M == 512 else "G24"


def main():
    obs = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, N)[:ROWS], dtype=np.int32)
    full2000 = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, N)[:2000], dtype=np.int32)
    par = params_for(M, N, full2000[:, 1:])          # keys of the 2 000-row slice the oracle test uses as well
    t = time.time()
    r = ref.estep(par["pi"], par["T"], par["keys"], par["E"], obs)
    dt = time.time() - t
    keys = [tuple(int(x) for x in k) for k in par["keys"]]
    gs = np.zeros((len(keys), M)); have = np.zeros(len(keys), dtype=bool)
    for i, k in enumerate(keys):
        if k in r["gamma_sums"]:
            gs[i] = r["gamma_sums"][k]; have[i] = True
    xs = r["xisum"]
    out = dict(a=par["a"], s=par["s"], hs=par["hs"], theta=par["theta"], rho=par["rho"], alpha=par["alpha"], pol=par["pol"], n=N,
               keys=par["keys"], rows=len(obs), crc=synth.contig_crc(obs), loglik=r["loglik"], q=r["q"],
               xisum_rowsum=xs.sum(axis=1), xisum_colsum=xs.sum(axis=0), xisum_diag=np.diag(xs).copy(), xisum_total=xs.sum(),
               xisum_offdiag_max=float((xs - np.diag(np.diag(xs))).max()), gs=gs, gs_have=have, gamma0=r["gamma"][:, 0].copy(),
               pi=par["pi"], T_diag=np.diag(par["T"]).copy(), T_rowsum=par["T"].sum(axis=1), E=par["E"], ref_seconds=dt)
    path = os.path.join(HERE, f"{TAG}_M{M}_n{N}_{ROWS}rows.npz")
    np.savez_compressed(path, **out)
    print(f"{TAG}: L={len(obs)} M={M} K={len(keys)} loglik={r['loglik']!r} in {dt:.1f} s -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
