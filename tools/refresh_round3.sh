#!/bin/bash
# round 3, final pass: GPU tests, smoke, kernel trace + PMC passes + SQ counters of the default bench, every bench line.  usage: tools/refresh_round3.sh <tag>
t=${1:-r03z}
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${t}_gputests.log 2>&1 || { tail -30 gpurun_out/${t}_gputests.log; exit 1; }
tail -2 gpurun_out/${t}_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${t}_smoke.log 2>&1 || { tail -20 gpurun_out/${t}_smoke.log; exit 1; }
tail -1 gpurun_out/${t}_smoke.log
bash tools/profile_round.sh ${t} > gpurun_out/${t}_profile.log 2>&1 || { tail -20 gpurun_out/${t}_profile.log; exit 1; }
echo profile done
bash tools/sq_profile.sh ${t}_sq --no-cpu-baseline --no-like-for-like --steps 4 > gpurun_out/${t}_sq.log 2>&1 || { tail -20 gpurun_out/${t}_sq.log; exit 1; }
echo sq done
bash tools/bench_all.sh ${t}
