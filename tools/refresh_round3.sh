#!/bin/bash
# round 3, second profile pass (after the rank compaction / lsdg_pops2 / reciprocal table): GPU tests, kernel trace + PMC passes + SQ counters of the default bench, every bench line
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03x_gputests.log 2>&1 || { tail -30 gpurun_out/r03x_gputests.log; exit 1; }
tail -2 gpurun_out/r03x_gputests.log
bash tools/profile_round.sh r03x > gpurun_out/r03x_profile.log 2>&1 || { tail -20 gpurun_out/r03x_profile.log; exit 1; }
echo profile done
bash tools/sq_profile.sh r03x_sq --no-cpu-baseline --no-like-for-like --steps 4 > gpurun_out/r03x_sq.log 2>&1 || { tail -20 gpurun_out/r03x_sq.log; exit 1; }
echo sq done
bash tools/bench_all.sh r03x
