#!/bin/bash
# every bench line of a round (run on the GPU box): tools/bench_all.sh <tag>   ->  gpurun_out/<tag>_bench_*.json
set -e
t=$1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${t}_bench_lines.json 2> gpurun_out/${t}_bench_lines.err
python bench.py --workload orb > gpurun_out/${t}_bench_orb.json 2> gpurun_out/${t}_bench_orb.err
python bench.py --workload dropin > gpurun_out/${t}_bench_dropin.json 2> gpurun_out/${t}_bench_dropin.err
python bench.py --workload tracking > gpurun_out/${t}_bench_tracking.json 2> gpurun_out/${t}_bench_tracking.err
python bench.py --host-io --batch 12288 --no-cpu-baseline > gpurun_out/${t}_bench_b12288io.json 2> gpurun_out/${t}_bench_b12288io.err
python bench.py --host-io --no-cpu-baseline > gpurun_out/${t}_bench_b6144io.json 2> gpurun_out/${t}_bench_b6144io.err
python bench.py --batch 6144 --no-cpu-baseline > gpurun_out/${t}_bench_lines_b6144.json 2> gpurun_out/${t}_bench_lines_b6144.err
python bench.py --batch 32 --host-io --no-cpu-baseline > gpurun_out/${t}_bench_b32io.json 2> gpurun_out/${t}_bench_b32io.err
for f in lines lines_b6144 orb dropin tracking b12288io b6144io b32io; do python - gpurun_out/${t}_bench_$f.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('_bench_')[1], d['value'], d['unit'], d['ms_per_step'], d.get('parity_checked_frames'), d.get('roofline', {}).get('frac'))
P
done
