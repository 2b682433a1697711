#!/bin/bash
# usage: tools/ab.sh "<label>|<env assignments>|<lib or ->|<bench args>" ...   (runs on the GPU box; prints one summary line per case)
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  IFS='|' read -r label envs lib args <<< "$spec"
  if [ "$lib" = "-" ]; then cmd="python bench.py"; else cmd="python tools/variant_bench.py $lib"; fi
  out=$(env $envs timeout -k 10 240 $cmd $args --cpu-seconds 0 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; p=d['pipeline']
    print('%-28s value=%8.1f ms=%7.3f frac=%.4f kernel_ms=%.4f x%d cands=%d hits=%d' % ('$label', d['value'], d['ms_per_step'], r['frac'], r['kernel_ms'], r['launches_per_step'], p['candidates'], p['raw_hits']))
except Exception as e:
    print('$label', 'FAILED', e)
"
done
