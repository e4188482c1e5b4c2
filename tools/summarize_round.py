"""Condenses gpurun_out/<dir> (made by tools/profile_round.sh) into profiles/<tag>_*:
   <tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats summary
   <tag>_bench_under_rocprof.json the bench line of the traced run
   <tag>_pmc_summary.txt          per-kernel FETCH_SIZE / WRITE_SIZE (own passes), corrected as calibrated
   <tag>_pmc_calibration.txt      the calibration kernels (known byte counts) measured in the same session
and updates profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.
Usage: python tools/summarize_round.py <dir under gpurun_out> <tag> <workload: orb|lines> <batch>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1])
tag, workload, batch = sys.argv[2], sys.argv[3], int(sys.argv[4])
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


def counters(sub, ctr):
    p = one(f"{sub}/**/*counter_collection.csv")
    agg = collections.defaultdict(list)
    if p:
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == ctr:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


stats_p = one("trace/**/*kernel_stats.csv")
shutil.copy(stats_p, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench_line = [l for l in open(os.path.join(src, "trace_bench.json")) if l.startswith("{")]
if bench_line:
    open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(bench_line[-1])
stats = {r["Name"]: r for r in csv.DictReader(open(stats_p))}
fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")

# calibration: known byte counts, same session
cal_f, cal_w = counters("calib_fetch", "FETCH_SIZE"), counters("calib_write", "WRITE_SIZE")
known = {"calib_read_b8": 1 << 28, "calib_read_b32": 1 << 30, "calib_read_b128": 1 << 30, "calib_write_b32": 1 << 30, "calib_write_b128": 1 << 30}
f_corr, w_corr = 2.0, 1.0
with open(os.path.join(dst, f"{tag}_pmc_calibration.txt"), "w") as f:
    f.write("# tools/pmc_calib/pmc_calib under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (own passes); counters in KiB\n")
    f.write(f"{'kernel':20s} {'known_bytes':>12s} {'counter_KiB':>12s} {'bytes/counted':>14s}\n")
    ratios_f, ratios_w = [], []
    for k, nbytes in known.items():
        table = cal_f if "read" in k else cal_w
        name = next((n for n in table if n.startswith(k)), None)
        if name is None:
            continue
        ratio = nbytes / (table[name] * 1024.0)
        (ratios_f if "read" in k else ratios_w).append(ratio)
        f.write(f"{k:20s} {nbytes:12d} {table[name]:12.1f} {ratio:14.3f}\n")
    if ratios_f:
        f_corr = sum(ratios_f) / len(ratios_f)
    if ratios_w:
        w_corr = sum(ratios_w) / len(ratios_w)
    f.write(f"# correction factors used: FETCH_SIZE x {f_corr:.3f} (1, 4 and 16 B/lane streaming reads all count half), WRITE_SIZE x {w_corr:.3f}\n")

traffic = {}
with open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w") as f:
    f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only; bench.py --workload {workload}, {batch} frames per launch.\n")
    f.write(f"# Mean per launch.  Raw counters are KiB; corrected MB = raw KiB * 1024 * factor / 1e6 with the factors of {tag}_pmc_calibration.txt\n")
    f.write(f"# (FETCH x {f_corr:.2f}, WRITE x {w_corr:.2f}).  avg_us from the separate --kernel-trace --stats run ({tag}_kernel_stats.csv).\n")
    f.write(f"{'kernel':44s} {'calls':>6s} {'avg_us':>10s} {'FETCH_KiB':>12s} {'WRITE_KiB':>12s} {'read_MB':>10s} {'write_MB':>10s} {'GB/s':>8s}\n")
    for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        if not (name.startswith("k_") or name.startswith("void k_")):
            continue
        fk, wk = fetch.get(name, float("nan")), write.get(name, float("nan"))
        rb, wb = fk * 1024 * f_corr, wk * 1024 * w_corr
        us = float(r["AverageNs"]) / 1e3
        f.write(f"{name[:44]:44s} {r['Calls']:>6s} {us:10.1f} {fk:12.1f} {wk:12.1f} {rb / 1e6:10.1f} {wb / 1e6:10.1f} {(rb + wb) / us / 1e3:8.0f}\n")
        short = name.split("(")[0].replace("void ", "")
        traffic[short] = {"read_bytes": rb, "write_bytes": wb, "avg_us": us, "calls": int(r["Calls"])}
jp = os.path.join(dst, "pmc_traffic.json")
allj = json.load(open(jp)) if os.path.exists(jp) else {}
allj[workload] = {"tag": tag, "frames_per_launch": batch, "fetch_correction": f_corr, "write_correction": w_corr, "kernels": traffic}
json.dump(allj, open(jp, "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{tag}_pmc_summary.txt")).read())
print(open(os.path.join(dst, f"{tag}_pmc_calibration.txt")).read())
