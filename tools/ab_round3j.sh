#!/bin/bash
# round 3: parity tests, then the default bench (run after a change of k_lsd_grow4 that has no build switch)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r03v_tests.log 2>&1 || { tail -30 gpurun_out/r03v_tests.log; exit 1; }
tail -3 gpurun_out/r03v_tests.log
python bench.py --prepare-inputs
for rep in 1 2; do
timeout -k 10 400 python bench.py --no-cpu-baseline --no-like-for-like --steps 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], {k: round(v, 2) for k, v in d.get('stages_ms_per_step', {}).items() if k.startswith('line.')})"
done
