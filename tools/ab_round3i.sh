#!/bin/bash
# round 3: lsdg_pops2 (packed f32 decision arithmetic, no `fresh` flag) against lsdg_pops, A/B in one session, after the parity tests
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 tools/diag/pk_check.hip -o /tmp/pk_check 2>/dev/null && timeout -k 10 60 /tmp/pk_check || exit 1
timeout -k 10 900 python -m pytest tests/test_line_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r03u_tests.log 2>&1 || { tail -30 gpurun_out/r03u_tests.log; exit 1; }
tail -3 gpurun_out/r03u_tests.log
python bench.py --prepare-inputs
bash tools/ab_build.sh "--no-cpu-baseline --no-like-for-like --steps 8" "-DPSL_GROW_ASM_POPS=1" "-DPSL_GROW_ASM_POPS=2" > gpurun_out/r03u_ab_pops2.log 2>&1
cat gpurun_out/r03u_ab_pops2.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
