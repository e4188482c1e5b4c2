#!/bin/bash
# round 3: where k_fast_cells4's time goes - parts done twice (identical results): quick test (1), scores (2), NMS + raster output (3), tile load + clears (4)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 5" "-DPSL_FAST_DIAG=0" "-DPSL_FAST_DIAG=1" "-DPSL_FAST_DIAG=2" "-DPSL_FAST_DIAG=3" "-DPSL_FAST_DIAG=4" > gpurun_out/r03g_fast_parts_twice.log 2>&1
cat gpurun_out/r03g_fast_parts_twice.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
