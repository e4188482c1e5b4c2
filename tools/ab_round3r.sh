#!/bin/bash
# round 3: lanes per NFA scan group (8 / 16 / 32), parity of the non-default sizes first
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
for f in "-DPSL_NFA_GL=8" "-DPSL_NFA_GL=32"; do
  PSLFE_EXTRA_FLAGS="$f" python psl-slam_amd/build.py --force > /dev/null 2>&1 || exit 1
  timeout -k 10 600 python -m pytest tests/test_line_gpu.py -x -q > gpurun_out/gl.log 2>&1 || { echo "$f: TESTS FAILED"; tail -20 gpurun_out/gl.log; exit 1; }
  echo "$f: $(tail -1 gpurun_out/gl.log)"
done
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 6" "-DPSL_NFA_GL=16" "-DPSL_NFA_GL=8" "-DPSL_NFA_GL=32" > gpurun_out/r03z_ab_nfa_gl.log 2>&1
cat gpurun_out/r03z_ab_nfa_gl.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
