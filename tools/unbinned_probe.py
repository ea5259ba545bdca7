#!/usr/bin/env python
"""Un-binned rows (the input of `smc++ posterior`) at 64 < M <= 256, three routes: the default (dense streamed chains, eigensystem
statistics, per-row gammas from eigen-power pieces + scan steps), the rows cut into 64-position pieces (SMCPP_SPLIT_SPANS=2: scan chains
walking every base pair, eigen-free statistics, per-row gammas by scan steps), and rounds 1-5 (SMCPP_GAMMA_PIECES=0: the scalar
eigensystem kernel for the per-row gammas).   python tools/unbinned_probe.py   (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smcpp_amd import _engine, _smcpp, synth  # noqa: E402
from smcpp_amd.model import PiecewiseModel  # noqa: E402

rows = int(os.environ.get("PROBE_ROWS", 100000))
GAMMA = os.environ.get("PROBE_GAMMA", "1") != "0"
for M in (96, 128, 256):
    res = {}
    for mode in ("2", "old", None):
        _engine.set_option("SMCPP_SPLIT_SPANS", "2" if mode == "2" else None)
        _engine.set_option("SMCPP_GAMMA_PIECES", "0" if mode == "old" else None)
        obs = np.ascontiguousarray(synth.synth_posterior_contig(rows, 8, seed=7), dtype=np.int32)
        a, s = synth.model_pieces()
        t0 = time.perf_counter()
        im = _smcpp.PyOnePopInferenceManager(8, [obs], synth.hidden_states(M), ("pop1",), 0.5)
        im.theta = 2e-4; im.rho = 6e-5; im.alpha = 1.0
        im.save_gamma = GAMMA
        model = PiecewiseModel(a, s, 1e4, "pop1")
        im.model = model; im.E_step(); ll = im.loglik()
        t1 = time.perf_counter()
        im.model = model; im.E_step(); ll = im.loglik()
        t2 = time.perf_counter()
        arg = np.asarray(im.gamma_argmax(0)) if GAMMA else np.zeros(1)
        t3 = time.perf_counter()
        p = im.describe()["plan"]
        res[mode] = (ll, arg)
        print(f"M = {M}, {rows} un-binned rows ({int(obs[:, 0].sum())} bp), pieces cut: {p['long_rows_cut']} ({p['rows']} rows): construction + first "
              f"E-step {t1 - t0:.2f} s, E-step {1e3 * (t2 - t1):.1f} ms, argmax {1e3 * (t3 - t2):.1f} ms, family {p['chain_family']}, passes "
              f"{p['passes_launched']}, per-row gamma {p['per_row_gamma']}, loglik {ll:.6f}", flush=True)
    (l1, a1), (l0, a0), (l2, a2) = res["2"], res[None], res["old"]
    print(f"   loglik rel diff (cut vs default) {abs(l1 - l0) / abs(l0):.2e}, decoded index differs on {int((a1 != a0).sum())} (cut vs default) / "
          f"{int((a2 != a0).sum())} (eigensystem kernel vs default) of {len(a0)} columns", flush=True)
_engine.set_option("SMCPP_SPLIT_SPANS", None)
_engine.set_option("SMCPP_GAMMA_PIECES", None)
