#!/bin/bash
# round 3: frames per workgroup of k_lsd_nfa_series (series of one length class of PSL_NFA_FG frames share a workgroup)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_NFA_FG=16" "-DPSL_NFA_FG=8" "-DPSL_NFA_FG=32" "-DPSL_NFA_FG=64" > gpurun_out/r03z_ab_nfa_fg.log 2>&1
cat gpurun_out/r03z_ab_nfa_fg.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
