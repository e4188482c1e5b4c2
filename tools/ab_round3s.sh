#!/bin/bash
# round 3: workgroups per frame and register bound of k_lsd_nfa_count (second sweep: fewer workgroups, 8 waves)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_NFA_COUNT_WGS=4 -DPSL_NFA_COUNT_WAVES=8" "-DPSL_NFA_COUNT_WGS=3 -DPSL_NFA_COUNT_WAVES=8" "-DPSL_NFA_COUNT_WGS=2 -DPSL_NFA_COUNT_WAVES=8" "-DPSL_NFA_COUNT_WGS=2 -DPSL_NFA_COUNT_WAVES=4" "-DPSL_NFA_COUNT_WGS=1 -DPSL_NFA_COUNT_WAVES=8" > gpurun_out/r03z_ab_nfa_grid2.log 2>&1
cat gpurun_out/r03z_ab_nfa_grid2.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
