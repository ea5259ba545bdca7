"""Golden G16: the graded configurations at FULL SIZE from the COMPILED REFERENCE (oracle/_ref: the reference's own
src/hmm.cpp, transition_bundle.cpp ... built where they lie by oracle/Makefile; build container only):

    make -C oracle ref && python tests/golden/make_golden_fullsize.py [--jobs 6] [--only headline,c2,...]

One `HMM::Estep` (src/hmm.cpp:45-153) per contig, single thread per contig exactly as the reference runs it
(src/inference_manager.cpp:89-94), on the rows `smcpp_amd.synth` generates and the parameters of the committed fixtures:

  headline   contigs 0..7 (the eight ranks' contigs of the weak-scaling bench), 100 Mbp, M = 64, n = 20
             contig 0: loglik, Q, xisum, gamma sums, gamma[:, 0]; contigs 1..7: loglik + Q
  c2         contig 0, 100 Mbp, M = 32, n = 10: the same set
  c3         the 22 contigs of the whole genome (2 872 Mbp, 6.76 M rows), M = 64, n = 20: per-contig loglik + Q, and the
             statistics summed over contigs in contig order (what `Q` sees, inference_manager.cpp:116-126)
  c4         contig 0, two populations, 100 Mbp, M = 48, on G13's parameters: the full set
  c5         the first 25 000 rows of the M = 256, n = 50 contig: the full set

Data only: scalars and M x M / K x M arrays of the reference's outputs (each file < 1 MB); the rows are NOT stored — the
generator is deterministic and `crc` pins them.  Results are merged into tests/golden/G16_fullsize_<name>.npz.
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

from smcpp_amd import synth  # noqa: E402

C5_ROWS = 25_000


def _params(name):
    if name in ("headline", "c3"):
        return dict(np.load(os.path.join(HERE, "params_M64_n20.npz")))
    if name == "c2":
        return dict(np.load(os.path.join(HERE, "params_M32_n10.npz")))
    if name == "c5":
        return dict(np.load(os.path.join(HERE, "params_M256_n50.npz")))
    if name == "c4":
        return dict(np.load(os.path.join(HERE, "G13_c4_params.npz")))
    raise KeyError(name)


def _rows(name, idx):
    if name == "headline":
        return synth.synth_contig(idx, 100_000_000, 20)
    if name == "c2":
        return synth.synth_contig(idx, 100_000_000, 10)
    if name == "c3":
        return synth.synth_contig(idx, int(synth.C3_LENGTHS_MBP[idx] * 1e6), 20)
    if name == "c4":
        return synth.synth_contig_twopop(idx, 100_000_000, 10, 10)
    if name == "c5":
        return np.ascontiguousarray(synth.synth_contig(idx, 100_000_000, 50)[:C5_ROWS])
    raise KeyError(name)


def run_task(task):
    name, idx = task
    from oracle import ref
    g = _params(name)
    obs = np.ascontiguousarray(_rows(name, idx), dtype=np.int32)
    known = {tuple(int(x) for x in k) for k in g["keys"]}
    present = {tuple(r) for r in np.unique(obs[:, 1:], axis=0).tolist()}
    assert present <= known, (name, idx, sorted(present - known)[:5])
    t = time.time()
    r = ref.estep(g["pi"], g["T"], g["keys"], g["E"], obs)
    dt = time.time() - t
    keys = [tuple(int(x) for x in k) for k in g["keys"]]
    gs = np.zeros((len(keys), len(g["pi"])))
    have = np.zeros(len(keys), dtype=bool)
    for i, k in enumerate(keys):
        if k in r["gamma_sums"]:
            gs[i] = r["gamma_sums"][k]; have[i] = True
    print(f"[{name} {idx}] rows {len(obs)} positions {int(obs[:, 0].sum())} loglik {r['loglik']!r} in {dt:.1f} s", flush=True)
    return dict(name=name, idx=idx, rows=len(obs), positions=int(obs[:, 0].sum()), crc=synth.contig_crc(obs),
                loglik=r["loglik"], q=r["q"], xisum=r["xisum"], gs=gs, gs_have=have, gamma0=r["gamma"][:, 0].copy(),
                seconds=dt)


TASKS = {
    "headline": [("headline", i) for i in range(8)],
    "c2": [("c2", 0)],
    "c3": [("c3", i) for i in range(22)],
    "c4": [("c4", 0)],
    "c5": [("c5", 0)],
}


def write(name, res):
    res = sorted(res, key=lambda r: r["idx"])
    g = _params(name)
    out = dict(keys=g["keys"], contig=np.array([r["idx"] for r in res]), rows=np.array([r["rows"] for r in res]),
               positions=np.array([r["positions"] for r in res]), crc=np.array([r["crc"] for r in res], dtype=np.int64),
               loglik=np.array([r["loglik"] for r in res]), q=np.array([r["q"] for r in res]),
               ref_seconds=np.array([r["seconds"] for r in res]))
    if name == "c3":
        # the statistics Q sums over the contigs (inference_manager.cpp:116-126), accumulated in contig order
        xs = np.zeros_like(res[0]["xisum"]); gs = np.zeros_like(res[0]["gs"]); g0 = np.zeros_like(res[0]["gamma0"])
        for r in res:
            xs += r["xisum"]; gs += r["gs"]; g0 += r["gamma0"]
        out.update(xisum_total=xs, gs_total=gs, gamma0_total=g0,
                   xisum_trace=np.array([np.trace(r["xisum"]) for r in res]),
                   xisum_sum=np.array([r["xisum"].sum() for r in res]))
    else:
        r0 = res[0]
        out.update(xisum=r0["xisum"], gs=r0["gs"], gs_have=r0["gs_have"], gamma0=r0["gamma0"])
    path = os.path.join(HERE, f"G16_fullsize_{name}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--only", default="headline,c2,c3,c4,c5")
    a = ap.parse_args()
    names = a.only.split(",")
    tasks = [t for nme in names for t in TASKS[nme]]
    # longest first: the c5 slice and the big c3 contigs
    cost = lambda t: (10.0 if t[0] == "c5" else synth.C3_LENGTHS_MBP[t[1]] / 100.0 if t[0] == "c3" else 1.0)  # noqa: E731
    tasks.sort(key=cost, reverse=True)
    res = {}
    with mp.get_context("spawn").Pool(a.jobs) as pool:
        for r in pool.imap_unordered(run_task, tasks):
            res.setdefault(r["name"], []).append(r)
            if len(res[r["name"]]) == len(TASKS[r["name"]]):
                write(r["name"], res[r["name"]])


if __name__ == "__main__":
    main()
