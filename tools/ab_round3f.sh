#!/bin/bash
# round 3: reduce_region_radius compacted by rank (three coalesced passes per radius step) against the entry-by-entry walk, A/B in one session, after the parity tests
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r03r_tests.log 2>&1 || { tail -30 gpurun_out/r03r_tests.log; exit 1; }
tail -3 gpurun_out/r03r_tests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 8" "-DPSL_REDUCE_SERIAL=1" "-DPSL_REDUCE_SERIAL=0" > gpurun_out/r03r_ab_reduce.log 2>&1
cat gpurun_out/r03r_ab_reduce.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
