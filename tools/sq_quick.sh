#!/bin/bash
# one SQ counter pass over tools/bench_lines.py (run on the GPU box): tools/sq_quick.sh <outdir under gpurun_out> <B> <style>
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
  -d $out/sq1 -o r --output-format csv -- python3 $repo/tools/bench_lines.py "$@" > $out/sq1.log 2>&1
python3 - "$out" <<'P'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + '/sq1/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:28]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k in acc:
    if 'grow' in k or 'grad' in k or 'scale' in k:
        print(k, n[k], {c: '%.4g' % (v / max(n[k], 1)) for c, v in acc[k].items()})
P
