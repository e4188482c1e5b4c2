#!/bin/bash
# round 3: the `used` bits of a few-frames launch in LDS (k_lsd_grow4<3, 1>) against the map in memory: parity tests, then one frame at a time at 640x480 and 1280x960
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_dropin_gpu.py tests/test_glue_gpu.py tests/test_robustness_gpu.py -x -q > gpurun_out/r03z_lu_tests.log 2>&1 || { tail -30 gpurun_out/r03z_lu_tests.log; exit 1; }
tail -2 gpurun_out/r03z_lu_tests.log
for rep in 1 2; do
for f in "-DPSL_GROW_LDS_USED=0" "-DPSL_GROW_LDS_USED=1"; do
  PSLFE_EXTRA_FLAGS="$f" python psl-slam_amd/build.py --force > /dev/null 2>&1 || exit 1
  for wl in dropin tracking; do
    timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f $wl', d['ms_per_step'], {k: round(v, 3) for k, v in d.get('calls_ms_mean', {}).items() if 'LINE' in k})" || exit 1
  done
  timeout -k 10 300 python bench.py --batch 32 --host-io --no-cpu-baseline --no-like-for-like --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f b32io', round(d['value'],1), d['ms_per_step'])" || exit 1
done
done > gpurun_out/r03z_ab_lds_used.log 2>&1
cat gpurun_out/r03z_ab_lds_used.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
