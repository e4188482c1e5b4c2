#!/bin/bash
# k_lsd_grow4 with / without helper waves at small batch sizes (run on the GPU box; builds the library in place)
for hf in 0 64; do
  PSLFE_EXTRA_FLAGS=-DPSL_GROW_HELPER_FRAMES=$hf python psl-slam_amd/build.py --force > /dev/null 2>&1 || exit 1
  for b in 1 8 16 32 64; do
    echo "== helper frames <= $hf, B = $b"
    timeout -k 10 200 python tools/bench_lines.py $b struct | grep "grow\|B=" || exit 1
  done
done
