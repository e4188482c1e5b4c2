#!/bin/bash
# round 3: frames per sub-batch of the scale / gradient launches
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_LSD_SUBBATCH=2048" "-DPSL_LSD_SUBBATCH=1024" "-DPSL_LSD_SUBBATCH=4096" "-DPSL_LSD_SUBBATCH=6144" > gpurun_out/r03z_ab_subbatch.log 2>&1
cat gpurun_out/r03z_ab_subbatch.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
