#!/usr/bin/env python
"""Copy the summaries of a tools/final_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked):
    python tools/collect_profiles.py <tag>"""
import glob
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
src = f"gpurun_out/{tag}"
subprocess.check_call([sys.executable, "tools/summarize_pmc.py", src, f"profiles/{tag}"])
for f in sorted(glob.glob(f"{src}/bench_*.log")):
    name = os.path.basename(f)
    if name != "bench_default.log":
        shutil.copy(f, f"profiles/{tag}_{name}")
for w in ("c3", "c5", "posterior", "posterior64", "posterior128", "qgrad"):
    ks = glob.glob(f"{src}/stats_{w}/*/*kernel_stats.csv")
    if ks:
        shutil.copy(ks[0], f"profiles/{tag}_{w}_kernel_stats.csv")
for f in glob.glob(f"{src}/{tag}_*_sq_counters.json"):
    shutil.copy(f, "profiles/" + os.path.basename(f))
for name in ("shard_probe.log", "dpp_lab.log", "sync_lab.log", "host_trace.log", "exchange_cost.log", "c4_host_timing.log", "gamma_scan_probe.log", "unbinned_probe.log",
             "poison_probe.log", "c4_watch.txt", "gpu_suite.log",
             "stats_timeline_headline.txt", "stats_timeline_c5.txt"):
    if os.path.exists(f"{src}/{name}"):
        shutil.copy(f"{src}/{name}", f"profiles/{tag}_{name}")
print(sorted(os.path.basename(p) for p in glob.glob(f"profiles/{tag}_*")))
