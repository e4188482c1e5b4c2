#!/bin/bash
# every bench line of a round (run on the GPU box): tools/bench_all.sh <tag>   ->  gpurun_out/<tag>_bench_*.json
set -e
t=$1
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python bench.py "$@" > gpurun_out/${t}_bench_$name.json 2> gpurun_out/${t}_bench_$name.err || { tail -5 gpurun_out/${t}_bench_$name.err; exit 1; }; }
run lines
run orb --workload orb
run struct --scene struct --no-cpu-baseline --no-like-for-like
run dropin --workload dropin
run dropin_K32 --workload dropin --lookahead 32 --no-cpu-baseline --steps 256
run tracking --workload tracking
run tracking_K8 --workload tracking --lookahead 8 --no-cpu-baseline --steps 256
run tracking_K32 --workload tracking --lookahead 32 --no-cpu-baseline --steps 256
run tracking_K128 --workload tracking --lookahead 128 --no-cpu-baseline --steps 256
run b12288io --host-io --batch 12288 --no-cpu-baseline --no-like-for-like
run b6144io --host-io --batch 6144 --no-cpu-baseline --no-like-for-like
run b32io --batch 32 --host-io --no-cpu-baseline --no-like-for-like
for f in lines orb struct dropin dropin_K32 tracking tracking_K8 tracking_K32 tracking_K128 b12288io b6144io b32io; do python - gpurun_out/${t}_bench_$f.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('_bench_')[1], d['value'], d['unit'], d['ms_per_step'], d.get('parity_checked_frames'), d.get('roofline', {}).get('frac'))
P
done
