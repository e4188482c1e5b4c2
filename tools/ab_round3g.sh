#!/bin/bash
# round 3: cost of k_lsd_grow4's parts by doing them twice (identical results): region2rect (4), the refinement's own loops (5)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_GROW_DIAG=0" "-DPSL_GROW_DIAG=4" "-DPSL_GROW_DIAG=5" > gpurun_out/r03s_grow_twice.log 2>&1
cat gpurun_out/r03s_grow_twice.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
