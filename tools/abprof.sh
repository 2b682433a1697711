#!/bin/bash
# usage: tools/abprof.sh "<label>|<env>|<lib or ->|<bench args>|<kernel substring>" ...  : per-kernel avg time through rocprofv3
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  IFS='|' read -r label envs lib args kern <<< "$spec"
  for e in $envs; do export "$e"; done
  O=$GRAFT_REPO_ROOT/gpurun_out/abprof
  rm -rf $O; mkdir -p $O
  if [ "$lib" = "-" ]; then prog="$GRAFT_REPO_ROOT/bench.py"; else prog="$GRAFT_REPO_ROOT/tools/variant_bench.py $GRAFT_REPO_ROOT/$lib"; fi
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $prog --cpu-seconds 0 $args > $O/log 2>&1)
  echo -n "$label: "; python tools/kstats.py $O/trace/*/*kernel_stats.csv | grep "$kern" | awk '{printf "%s calls=%s avg_us=%s  ", $1, $2, $4}'; echo
  for e in $envs; do unset "${e%%=*}"; done
done
