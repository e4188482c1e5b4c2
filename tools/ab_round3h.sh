#!/bin/bash
# round 3: k_lsd_grow4 at 7 against 8 waves per SIMD after the rank compaction (8: 13 SGPR spill reloads per window round, 7: 2)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_GROW_WAVES=8" "-DPSL_GROW_WAVES=7" > gpurun_out/r03t_ab_waves.log 2>&1
cat gpurun_out/r03t_ab_waves.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
