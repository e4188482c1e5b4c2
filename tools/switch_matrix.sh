#!/bin/bash
# the documented build switches still build and still give the reference's results: tests/test_line_gpu.py under each of them (GPU box)
cd $GRAFT_REPO_ROOT
for f in "-DPSL_LSD_RING=128" "-DPSL_REDUCE_SERIAL=1" "-DPSL_GROW_ASM_POPS=1" "-DPSL_GROW_ASM_POPS=0" "-DPSL_FRAME_ORDER=0" "-DPSL_GROW_WAVES=7" "-DPSL_NFA_BYROW_T=16 -DPSL_NFA_BYROW_MIN=16"; do
  PSLFE_EXTRA_FLAGS="$f" python psl-slam_amd/build.py --force > /dev/null 2>&1 || { echo "$f: BUILD FAILED"; exit 1; }
  timeout -k 10 600 python -m pytest tests/test_line_gpu.py -x -q > gpurun_out/switch.log 2>&1 || { echo "$f: TESTS FAILED"; tail -30 gpurun_out/switch.log; exit 1; }
  echo "$f: $(tail -1 gpurun_out/switch.log)"
done
python psl-slam_amd/build.py --force > /dev/null 2>&1
