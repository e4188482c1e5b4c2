#!/bin/bash
# round 3: hand-written pop loop of k_lsd_grow4 (lsdg_pops) against the compiled C++ loop, A/B in one session, after the parity tests
cd $GRAFT_REPO_ROOT
python psl-slam_amd/build.py --force > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r03l_tests.log 2>&1 || { tail -20 gpurun_out/r03l_tests.log; exit 1; }
tail -3 gpurun_out/r03l_tests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 8" "-DPSL_GROW_ASM_POPS=0" "-DPSL_GROW_ASM_POPS=1" > gpurun_out/r03l_ab_pops.log 2>&1
cat gpurun_out/r03l_ab_pops.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
python bench.py --workload dropin --no-cpu-baseline > gpurun_out/r03l_dropin.json 2> gpurun_out/r03l_dropin.err
