#!/bin/bash
# round 3: frames heaviest-first in k_lsd_grow4 (A/B in one session) after the line parity tests
cd $GRAFT_REPO_ROOT
python psl-slam_amd/build.py --force > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_match_gpu.py -x -q > gpurun_out/r03i_tests.log 2>&1 || { tail -20 gpurun_out/r03i_tests.log; exit 1; }
tail -3 gpurun_out/r03i_tests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 8" "-DPSL_FRAME_ORDER=0" "-DPSL_FRAME_ORDER=1" > gpurun_out/r03i_ab_order.log 2>&1
cat gpurun_out/r03i_ab_order.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
python bench.py --scene struct --no-cpu-baseline --no-like-for-like --steps 8 > gpurun_out/r03i_struct.json 2> gpurun_out/r03i_struct.err
