#!/bin/bash
# round 3: FAST scores with packed 16-bit min / max (psl_fast_score_pol_pk) against one difference per instruction: ORB parity tests, then A/B
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_orb_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r03g_orbtests.log 2>&1 || { tail -30 gpurun_out/r03g_orbtests.log; exit 1; }
tail -2 gpurun_out/r03g_orbtests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 5" "-DPSL_FAST_SCORE_PK=0" "-DPSL_FAST_SCORE_PK=1" > gpurun_out/r03g_ab_fast_score_pk.log 2>&1
cat gpurun_out/r03g_ab_fast_score_pk.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
timeout -k 10 300 python bench.py --workload orb --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('orb workload', d['value'], d['ms_per_step'], d['parity_checked_frames'])"
