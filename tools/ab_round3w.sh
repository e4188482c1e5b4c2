#!/bin/bash
# round 3: per-frame fetches through pinned staging - the tests that use them, then the one-frame-at-a-time and look-ahead lines
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dropin_gpu.py tests/test_glue_gpu.py tests/test_match_gpu.py tests/test_robustness_gpu.py tests/test_prep_gpu.py -x -q > gpurun_out/r03g_tests.log 2>&1 || { tail -30 gpurun_out/r03g_tests.log; exit 1; }
tail -2 gpurun_out/r03g_tests.log
for wl in dropin tracking; do
  python bench.py --workload $wl --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl K1', d['ms_per_step'], {k: round(v, 3) for k, v in d.get('calls_ms_mean', {}).items()})"
done
bash tools/bench_k.sh r03g
python - <<'P'
import json
for f in ('dropin_K32','tracking_K32'):
    d=json.loads(open('gpurun_out/r03g_bench_%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], {k: round(v,3) for k,v in d['calls_ms_mean'].items()})
P
