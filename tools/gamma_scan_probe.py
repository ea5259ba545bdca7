#!/usr/bin/env python
"""Posterior decode of BINNED contigs (save_gamma E-step + argmax on the device) with the per-row posteriors of the span > 1 rows from
scan steps (round 6, default) and from eigensystems (SMCPP_GAMMA_SCAN=0):   python tools/gamma_scan_probe.py   (GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smcpp_amd import _engine, _smcpp, synth  # noqa: E402
from smcpp_amd.model import PiecewiseModel  # noqa: E402


def run(M, n, rows, scan, steps):
    _engine.set_option("SMCPP_GAMMA_SCAN", scan)
    obs = np.ascontiguousarray(synth.synth_contig(0, 100_000_000, n)[:rows], dtype=np.int32)
    a, s = synth.model_pieces()
    im = _smcpp.PyOnePopInferenceManager(n, [obs], synth.hidden_states(M), ("pop1",), 0.5)
    im.theta = synth.THETA; im.rho = synth.RHO; im.alpha = 1.0
    im.save_gamma = True
    model = PiecewiseModel(a, s, 1e4, "pop1")
    im.model = model; im.E_step()
    t0 = time.perf_counter()
    for _ in range(steps):
        im.model = model
        im.E_step()
        arg = im.gamma_argmax(0)
    dt = (time.perf_counter() - t0) / steps
    return dt, arg, im.describe()["plan"]["per_row_gamma"], im.loglik()


for M, n, rows, steps in ((64, 20, 235552, 5), (256, 50, 235552, 2), (512, 10, 60000, 2)):
    res = {}
    for scan in ("1", "0"):
        if M > 256 and scan == "0":
            continue
        dt, arg, how, ll = run(M, n, rows, scan, steps)
        res[scan] = arg
        print(f"M = {M}, {rows} rows: save_gamma E-step + device argmax {1e3 * dt:9.2f} ms  per-row gamma by {how}  loglik {ll:.6f}", flush=True)
    if "0" in res:
        print(f"   decoded index differs on {int((res['0'] != res['1']).sum())} of {len(res['1'])} columns")
_engine.set_option("SMCPP_GAMMA_SCAN", None)
