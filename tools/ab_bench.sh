#!/bin/bash
# usage: tools/ab_bench.sh "<bench args>" VAR=1 VAR2=1 ...   (first run = no variable)
args="$1"; shift
for v in "" "$@"; do
  echo "== ${v:-default}"
  env $v timeout -k 10 300 python bench.py $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], json.dumps(d.get('stages_ms_per_launch', {})))"
done
