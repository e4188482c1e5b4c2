#!/bin/bash
# the look-ahead lines of configs[4] / configs[2] (256 frames each): tools/bench_k.sh <tag>
set -e
t=$1
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python bench.py "$@" > gpurun_out/${t}_bench_$name.json 2> gpurun_out/${t}_bench_$name.err || { tail -5 gpurun_out/${t}_bench_$name.err; exit 1; }; }
run dropin_K32 --workload dropin --lookahead 32 --no-cpu-baseline --steps 256
run tracking_K8 --workload tracking --lookahead 8 --no-cpu-baseline --steps 256
run tracking_K32 --workload tracking --lookahead 32 --no-cpu-baseline --steps 256
run tracking_K128 --workload tracking --lookahead 128 --no-cpu-baseline --steps 256
for f in dropin_K32 tracking_K8 tracking_K32 tracking_K128; do python - gpurun_out/${t}_bench_$f.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('_bench_')[1], d['value'], d['unit'], d['ms_per_step'], d.get('parity_checked_frames'))
P
done
