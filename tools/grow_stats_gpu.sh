#!/bin/bash
# in-kernel counts and cycle stamps of k_lsd_grow4 for ONE frame (diagnostic build -DPSL_GROW_STATS; the pop loop is the compiled C++ one there): tools/grow_stats_gpu.sh [style]
cd $GRAFT_REPO_ROOT
PSLFE_EXTRA_FLAGS="-DPSL_GROW_STATS" python psl-slam_amd/build.py --force > /dev/null 2>&1 || exit 1
for style in ${1:-struct} sticks; do
for lu in 1; do
python - $style <<'P'
import sys, numpy as np
sys.path[:0] = ['.', 'tools', 'tests']
import psl_slam_amd as P, synth_frames as sf
img = np.ascontiguousarray(sf.Scene(640, 480, sys.argv[1], 3).gray(2))
le = P.LINEextractor(1, 1.2, 200, 0.0)
for _ in range(3):
    seg = le.lsd_detect(img)
print(sys.argv[1], 'segments', len(seg), flush=True)
P
done
done
python psl-slam_amd/build.py --force > /dev/null 2>&1
