#!/bin/bash
# round 3: register bounds (waves per SIMD) of k_lsd_grad, k_lbd, k_lil_pair, k_line_good
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_GRAD_WAVES=1" "-DPSL_GRAD_WAVES=8" "-DPSL_LBD_WAVES=6" "-DPSL_LBD_WAVES=8" "-DPSL_PAIR_WAVES=8" "-DPSL_GOOD_WAVES=5" "-DPSL_GOOD_WAVES=6" "-DPSL_GOOD_WAVES=3" > gpurun_out/r03z_ab_waves_misc.log 2>&1
cat gpurun_out/r03z_ab_waves_misc.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
