#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of tools/pmc_sq_counters.sh:
    python tools/summarize_sq.py <dir with p1/ p2/ ...> <out.json> <workload>
Per kernel (template arguments kept, parameter list dropped): launches, and per counter the sum over all launches of the run
divided by the number of E-steps of the run (= launches of k_loglik_final), plus the derived ratios MI355X_MICROARCH.md gives
units for: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count QUAD-cycles, SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES cycles."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(src, out, workload):
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{src}/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    n_esteps = max([len(next(iter(v.values()))) for k, v in vals.items() if "k_loglik_final" in k] + [1])
    res = {"note": "rocprofv3 --kernel-trace --pmc <set> (one pass per set), python bench.py --no-cpu --workload %s --steps 3 --warmup 1; "
                   "per_step = sum over all launches of the kernel in the run / E-steps of the run" % workload,
           "workload": workload, "esteps_in_run": n_esteps, "kernels": {}}
    for k, cs in sorted(vals.items()):
        e = {"launches": max(len(v) for v in cs.values()), "per_step": {c: sum(v) / n_esteps for c, v in cs.items()}}
        ps = e["per_step"]
        d = {}
        if ps.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                if c in ps:
                    d[c + "_over_WAVE_CYCLES"] = ps[c] / ps["SQ_WAVE_CYCLES"]
            if ps.get("SQ_INSTS_VALU"):
                d["cycles_per_valu_instr_per_wave"] = 4.0 * ps["SQ_WAVE_CYCLES"] / ps["SQ_INSTS_VALU"]
        if ps.get("SQ_BUSY_CYCLES") and ps.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            d["MFMA_BUSY_over_SQ_BUSY"] = ps["SQ_VALU_MFMA_BUSY_CYCLES"] / ps["SQ_BUSY_CYCLES"]
        e["derived"] = d
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, "kernels:", len(res["kernels"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
