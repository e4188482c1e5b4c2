#!/bin/bash
# round 3: k_lbd's sample loop with fewer instructions (parity first), then the NFA scans' row-assignment thresholds
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r03z_tests.log 2>&1 || { tail -30 gpurun_out/r03z_tests.log; exit 1; }
tail -2 gpurun_out/r03z_tests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_NFA_BYROW_T=16" "-DPSL_NFA_BYROW_T=6" "-DPSL_NFA_BYROW_T=3" "-DPSL_NFA_BYROW_MIN=6" "-DPSL_NFA_BYROW_MIN=3" > gpurun_out/r03z_ab_byrow.log 2>&1
cat gpurun_out/r03z_ab_byrow.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
