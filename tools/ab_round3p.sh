#!/bin/bash
# round 3: waves per SIMD of k_lsd_nfa_series (122 VGPRs = 4 waves without a bound; its dependent f64 chains leave the vector unit 54 % idle)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_NFA_SERIES_WAVES=1" "-DPSL_NFA_SERIES_WAVES=5" "-DPSL_NFA_SERIES_WAVES=6" "-DPSL_NFA_SERIES_WAVES=8" > gpurun_out/r03z_ab_series_waves.log 2>&1
cat gpurun_out/r03z_ab_series_waves.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
