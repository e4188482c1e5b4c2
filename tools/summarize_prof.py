"""Condenses rocprofv3 output under gpurun_out/prof/{trace,fetch,write} into profiles/<tag>_*.
Usage: python tools/summarize_prof.py r01a"""
import collections
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1]
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "r01_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "trace_bench.json")):
    shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "trace", "r01_kernel_stats.csv")))}
pmc = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    p = os.path.join(src, kind, "r01_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    pmc[ctr] = {k: sum(v) / len(v) for k, v in agg.items()}
with open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), mean per launch.\n")
    f.write("# Units: KiB as rocprofv3 reports them. gfx950: FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x\n")
    f.write("# (MI355X_MICROARCH.md §HBM); these kernels use narrower accesses, so the figures are uncalibrated absolutes.\n")
    f.write(f"{'kernel':60s} {'avg_us(trace)':>14s} {'FETCH_KiB':>12s} {'WRITE_KiB':>12s}\n")
    for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        if not (name.startswith("k_") or name.startswith("void k_")):
            continue
        f.write(f"{name[:60]:60s} {float(r['AverageNs']) / 1e3:14.1f} {pmc.get('FETCH_SIZE', {}).get(name, float('nan')):12.1f} "
                f"{pmc.get('WRITE_SIZE', {}).get(name, float('nan')):12.1f}\n")
print(open(os.path.join(dst, f"{tag}_pmc_summary.txt")).read())
