#!/bin/bash
# HBM traffic of the LSD kernels over tools/bench_lines.py (run on the GPU box): FETCH_SIZE and WRITE_SIZE in their own passes.
# usage: tools/pmc_quick.sh <outdir under gpurun_out> <B> <style>
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/$c -o r --output-format csv -- python3 $repo/tools/bench_lines.py "$@" > $out/$c.log 2>&1
done
python3 - "$out" "$1" <<'P'
import csv, sys, glob, collections
out, B = sys.argv[1], int(sys.argv[2])
for c, f in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):   # KiB counters; FETCH_SIZE counts half on gfx950 (profiles/*_pmc_calibration.txt)
    p = glob.glob(out + '/' + c + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] == c: acc[r['Kernel_Name'][:24]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        if 'lsd' in k: print(c, k, 'MB per frame %.3f' % (sum(v) / len(v) * 1024 * f / B / 1e6))
P
