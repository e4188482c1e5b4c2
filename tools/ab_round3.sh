#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_line_gpu.py -x -q > gpurun_out/r03c_linetests.log 2>&1 || { tail -20 gpurun_out/r03c_linetests.log; exit 1; }
tail -3 gpurun_out/r03c_linetests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 8" "-DPSL_GROW_WAVES=6" "-DPSL_GROW_WAVES=8 -DPSL_LSD_RING=512" "-DPSL_GROW_WAVES=7 -DPSL_LSD_RING=512" > gpurun_out/r03c_ab_waves.log 2>&1
python psl-slam_amd/build.py --force > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-like-for-like --steps 8 --streams 2 > gpurun_out/r03c_streams2.json 2> gpurun_out/r03c_streams2.err
for K in 64 128; do python bench.py --workload tracking --lookahead $K --no-cpu-baseline --steps 128 > gpurun_out/r03c_tracking_K$K.json 2> gpurun_out/r03c_tracking_K$K.err; done
cat gpurun_out/r03c_ab_waves.log
