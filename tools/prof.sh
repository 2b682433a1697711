#!/bin/bash
# usage: tools/prof.sh <name> [env assignments ...] -- <bench args>   (on the GPU box; kernel stats -> gpurun_out/<name>_stats.txt)
name=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
for e in "${envs[@]}"; do export "$e"; done
O=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 "$@" > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/kstats.py $O/trace/*/*kernel_stats.csv > $GRAFT_REPO_ROOT/gpurun_out/${name}_stats.txt
rm -rf $O/trace
head -16 $GRAFT_REPO_ROOT/gpurun_out/${name}_stats.txt
