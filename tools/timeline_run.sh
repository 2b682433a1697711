#!/bin/bash
# usage (GPU box): tools/timeline_run.sh <label> <bench args...>: kernel timeline of the last bench step -> gpurun_out/timeline_<label>.txt
# (environment knobs such as HG_MAX_CHUNKS are read from the caller's environment: export them before the call)
label=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/tl_$label
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-extra --steps 2 --warmup 1 "$@" > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/timeline.py $O/trace/*/*kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/timeline_$label.txt
rm -rf $O/trace
