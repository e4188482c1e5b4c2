#!/bin/bash
# round 3: the ORB pipeline on a second stream beside the line pipeline, with the growing at 8 / 7 / 6 waves per SIMD (free wave slots for the other stream's kernels)
cd $GRAFT_REPO_ROOT
python bench.py --prepare-inputs
one() { f=$1; shift
  PSLFE_EXTRA_FLAGS="$f" python psl-slam_amd/build.py --force > /dev/null 2>&1 || exit 1
  echo "== $f $@"
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-like-for-like --steps 6 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], {k: round(v, 2) for k, v in d.get('stages_ms_per_step', {}).items() if k in ('line.lsd_grow', 'orb.fast', 'orb.blur', 'line.nfa_count')})" || exit 1
}
for rep in 1 2; do
one -DPSL_GROW_WAVES=8 --streams 1
one -DPSL_GROW_WAVES=8 --streams 2
one -DPSL_GROW_WAVES=7 --streams 2
one -DPSL_GROW_WAVES=6 --streams 2
done > gpurun_out/r03z_ab_streams.log 2>&1
cat gpurun_out/r03z_ab_streams.log
python psl-slam_amd/build.py --force > /dev/null 2>&1
