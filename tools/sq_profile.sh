#!/bin/bash
# SQ counter passes for the bench (run on the GPU box): tools/sq_profile.sh <outdir under gpurun_out> <bench args...>
# Counters only with --kernel-trace (no other trace domains), one pass per 8 SQ counters.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
  -d $out/sq1 -o r --output-format csv -- python3 $repo/bench.py "$@" > $out/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR \
  -d $out/sq2 -o r --output-format csv -- python3 $repo/bench.py "$@" > $out/sq2.log 2>&1
python3 $repo/tools/sq_summary.py $out
