#!/bin/bash
for w in 5 6 7 8; do
  PSLFE_EXTRA_FLAGS=-DPSL_GROW_WAVES=$w python psl-slam_amd/build.py --force > /dev/null 2>&1 || exit 1
  echo "== waves $w"
  timeout -k 10 200 python tools/bench_lines.py 12288 struct | grep "grow\|B=" || exit 1
done
