"""Goldens G19 / G20: the reference's POSTERIOR DECODE at full scale, from the COMPILED REFERENCE (oracle/_ref: the
reference's own src/hmm.cpp ... built where they lie by oracle/Makefile; build container only):

    make -C oracle ref && python tests/golden/make_golden_argmax.py [--jobs 4] [--only G19_headline,...]

`HMM::Estep` with `save_gamma` (src/hmm.cpp:141-150: gamma.col(ell) of every row), the per-column argmax that
`smc++ posterior` reports (smcpp/commands/posterior.py:98-111), on

  G19_headline      contig 0 of the headline (100 Mbp, 235 552 rows, M = 64, n = 20; fixture params_M64_n20.npz)
  G19_c2            contig 0 of config C2 (M = 32, n = 10; fixture params_M32_n10.npz)
  G20_posterior     the un-binned contig of `bench.py --workload posterior`   (10^6 rows, spans to 1e5, M = 32, n = 8)
  G20_posterior64   the un-binned contig of `bench.py --workload posterior64` (the same rows at M = 64)

Data only, each file < 1 MB: the argmax of EVERY column (uint8), the columns whose relative top-1 / top-2 margin is below
1e-3 with their margins (every other column has a margin above 1e-3), a strided sample of gamma columns (float32), the
reference's loglik / Q / xisum / gamma sums / gamma[:, 0], and - for G20, whose parameters are not a committed fixture -
the prepared parameters the reference ran on (pi, T, keys, E from `ref_prep` + the emission assembly of oracle/prep_oracle.py)
with the model (a, s, hs, theta, rho) they came from.  The rows are NOT stored: the generators are deterministic and `crc`
pins them.
"""
from __future__ import annotations

import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from smcpp_amd import synth  # noqa: E402

LOW_MARGIN = 1e-3
GAMMA_SAMPLE_COLS = 1024


def _case(name):
    """-> (params dict with pi/T/keys/E/..., rows, extra fields to store)"""
    if name == "G19_headline":
        return dict(np.load(os.path.join(HERE, "params_M64_n20.npz"))), synth.synth_contig(0, 100_000_000, 20), False
    if name == "G19_c2":
        return dict(np.load(os.path.join(HERE, "params_M32_n10.npz"))), synth.synth_contig(0, 100_000_000, 10), False
    if name in ("G20_posterior", "G20_posterior64"):
        from make_golden import params_for
        M = 32 if name == "G20_posterior" else 64
        obs = synth.synth_posterior_contig(1_000_000, 8, seed=7)
        # bench.py's parameters of these workloads: theta = 2e-4, rho = 6e-5, the default model pieces and hidden states
        return params_for(M, 8, obs[:, 1:], theta=1e-4 * 2, rho=6e-5), obs, True
    raise KeyError(name)


def run(name):
    from oracle import ref
    g, obs, store_params = _case(name)
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    t = time.time()
    r = ref.estep(g["pi"], g["T"], g["keys"], g["E"], obs, save_gamma=True)
    dt = time.time() - t
    gam = r["gamma"]                                    # [M, L + 1]
    M = gam.shape[0]
    arg = gam.argmax(axis=0).astype(np.uint8)
    top2 = np.partition(gam, M - 2, axis=0)[M - 2:]
    margin = (top2[1] - top2[0]) / np.maximum(top2[1], 1e-300)
    low = np.nonzero(margin < LOW_MARGIN)[0].astype(np.int32)
    stride = max(1, gam.shape[1] // GAMMA_SAMPLE_COLS)
    keys = [tuple(int(x) for x in k) for k in g["keys"]]
    gs = np.zeros((len(keys), M)); have = np.zeros(len(keys), dtype=bool)
    for i, k in enumerate(keys):
        if k in r["gamma_sums"]:
            gs[i] = r["gamma_sums"][k]; have[i] = True
    out = dict(rows=len(obs), positions=int(obs[:, 0].sum()), crc=synth.contig_crc(obs), keys=g["keys"],
               loglik=r["loglik"], q=r["q"], xisum=r["xisum"], gs=gs, gs_have=have, gamma0=gam[:, 0].copy(),
               gamma_argmax=arg, low_margin_cols=low, low_margin=margin[low].astype(np.float32), low_margin_below=LOW_MARGIN,
               min_margin=float(margin.min()), gamma_stride=stride, gamma_sub=gam[:, ::stride].astype(np.float32),
               ref_seconds=dt)
    if store_params:
        out.update(pi=g["pi"], T=g["T"], E=g["E"], hs=g["hs"], a=g["a"], s=g["s"], theta=g["theta"], rho=g["rho"],
                   alpha=g["alpha"], pol=g["pol"], n=g["n"])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: L={len(obs)} M={M} loglik={r['loglik']!r} min margin={margin.min():.3g} "
          f"columns below {LOW_MARGIN:g}: {len(low)} (below 1e-5: {int((margin < 1e-5).sum())}) in {dt:.1f} s "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--only", default="G20_posterior64,G19_headline,G20_posterior,G19_c2")
    a = ap.parse_args()
    names = a.only.split(",")
    with mp.get_context("spawn").Pool(min(a.jobs, len(names))) as pool:
        for nme in pool.imap_unordered(run, names):
            pass


if __name__ == "__main__":
    main()
