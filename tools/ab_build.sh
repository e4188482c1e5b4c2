#!/bin/bash
# A/B of compile-time switches in ONE session on ONE device (boxes differ by several per cent): tools/ab_build.sh "<bench args>" FLAG_A FLAG_B ...
# e.g. tools/ab_build.sh "--no-cpu-baseline" -DPSL_GROW_PERMUTE=0 -DPSL_GROW_PERMUTE=1     (run on the GPU box; builds the library in place)
args="$1"; shift
for rep in 1 2; do
for f in "$@"; do
  PSLFE_EXTRA_FLAGS="$f" python psl-slam_amd/build.py --force > /dev/null 2>&1 || exit 1
  echo "== $f (run $rep)"
  timeout -k 10 400 python bench.py $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], {k: round(v, 2) for k, v in d.get('stages_ms_per_step', {}).items() if k in ('orb.fast', 'orb.blur', 'orb.describe')})" || exit 1
done
done
