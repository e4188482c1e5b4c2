#!/bin/bash
# round 3: wide decision masks with an acceptance budget in k_lsd_grow4's pop loop (lsdg_pops_b) against lsdg_pops, A/B in one session, after the parity tests
cd $GRAFT_REPO_ROOT
python psl-slam_amd/build.py --force > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r03o_tests.log 2>&1 || { tail -20 gpurun_out/r03o_tests.log; exit 1; }
tail -3 gpurun_out/r03o_tests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 8" "-DPSL_GROW_BUDGET=0" "-DPSL_GROW_BUDGET=1" > gpurun_out/r03o_ab_budget.log 2>&1
cat gpurun_out/r03o_ab_budget.log
