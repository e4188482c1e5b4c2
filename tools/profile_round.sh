#!/bin/bash
# Round profile on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes
# (counters only with --kernel-trace).  usage: tools/profile_round.sh <subdir of gpurun_out> <bench args...>
set -e
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the input frames are generated (worker pool) and cached by an UNPROFILED run; under the profiler bench.py starts no process at all
# (no pool, no CPU baseline, no device probe): the preload has initialised the GPU before the program starts
# every profiled pass runs the headline workload only (--no-like-for-like: the extra 6144-frame steps would dilute the per-kernel averages)
python3 $repo/bench.py "$@" --prepare-inputs
rocprofv3 --kernel-trace --stats -d $out/trace -o r --output-format csv -- python3 $repo/bench.py "$@" --no-cpu-baseline --no-like-for-like > $out/trace_bench.json 2> $out/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o r --output-format csv -- python3 $repo/bench.py "$@" --no-cpu-baseline --no-like-for-like > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o r --output-format csv -- python3 $repo/bench.py "$@" --no-cpu-baseline --no-like-for-like > $out/write.log 2>&1
if [ -x $repo/tools/pmc_calib/pmc_calib ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/calib_fetch -o r --output-format csv -- $repo/tools/pmc_calib/pmc_calib > $out/calib_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/calib_write -o r --output-format csv -- $repo/tools/pmc_calib/pmc_calib > $out/calib_write.log 2>&1
fi
ls $out
