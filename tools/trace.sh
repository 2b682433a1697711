#!/bin/bash
# usage: tools/trace.sh <name> [env ...] -- <bench args>  -> gpurun_out/<name>_timeline.txt (kernel timeline of the last step)
name=$1; shift
while [ "$1" != "--" ]; do export "$1"; shift; done
shift
O=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 "$@" > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/timeline.py $O/trace/*/*kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/${name}_timeline.txt
rm -rf $O/trace
cat $GRAFT_REPO_ROOT/gpurun_out/${name}_timeline.txt
