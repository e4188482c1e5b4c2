#!/bin/bash
# round 3: where k_lsd_grow4's time goes - diagnostic builds that leave parts of the chain out (results are NOT the reference's: parity check off)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6 --parity-frames 0" "-DPSL_GROW_DIAG=0" "-DPSL_GROW_DIAG=1" "-DPSL_GROW_DIAG=2" "-DPSL_GROW_DIAG=3" > gpurun_out/r03q_grow_parts.log 2>&1
cat gpurun_out/r03q_grow_parts.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
