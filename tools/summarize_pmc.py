#!/usr/bin/env python
"""Summarise the rocprofv3 outputs of tools/profile_round.sh into the files kept under profiles/:
    python tools/summarize_pmc.py gpurun_out/<tag> profiles/<tag>
writes <prefix>_kernel_stats.csv (copy), <prefix>_bench_default.log (copy) and <prefix>_hbm_traffic_pmc.json
(per kernel: max / median FETCH_SIZE and WRITE_SIZE in KB per launch, as rocprofv3 reports them)."""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict

import numpy as np


def main(src, prefix):
    ks = glob.glob(f"{src}/stats/*/*kernel_stats.csv")
    if ks:
        shutil.copy(ks[0], f"{prefix}_kernel_stats.csv")
    shutil.copy(f"{src}/bench_default.log", f"{prefix}_bench_default.log")
    vals = defaultdict(lambda: defaultdict(list))
    for d in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0]
                vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --no-cpu "
                   "--steps 2 --warmup 1, headline workload; KB per launch as rocprofv3 reports them (max = a pass that "
                   "re-ran every chunk; converged check passes write nothing). MI355X_MICROARCH.md: FETCH_SIZE "
                   "under-reports wide coalesced reads by 2x on gfx950, so bytes_per_step = (2 x FETCH_SIZE + WRITE_SIZE) "
                   "x 1024 summed over every launch of the kernel, divided by the number of E-steps of the run (= launches "
                   "of k_loglik_final).",
           "kernels": {}}
    n_esteps = max([len(v.get("FETCH_SIZE", [])) for k, v in vals.items() if "k_loglik_final" in k] + [1])
    out["esteps_in_run"] = n_esteps
    for k, cs in vals.items():
        e = {}
        tot = 0.0
        for c, v in cs.items():
            v = np.array(v)
            e[f"{c}_KB_max"] = float(v.max())
            e[f"{c}_KB_median"] = float(np.median(v))
            e[f"{c}_KB_sum"] = float(v.sum())
            e["launches"] = int(len(v))
            tot += (2.0 if c == "FETCH_SIZE" else 1.0) * float(v.sum()) * 1024.0
        e["bytes_per_step"] = tot / n_esteps
        out["kernels"][k] = e
    json.dump(out, open(f"{prefix}_hbm_traffic_pmc.json", "w"), indent=1)
    print("wrote", prefix)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
