#!/bin/bash
# usage (GPU box): tools/kstat_ab.sh <label> <lib or -> <bench args...>   -> per-kernel averages (rocprofv3 --kernel-trace --stats) of one bench run
label=$1; lib=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/kstat_$label
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$lib" = "-" ]; then prog="$GRAFT_REPO_ROOT/bench.py"; else prog="$GRAFT_REPO_ROOT/tools/variant_bench.py $GRAFT_REPO_ROOT/$lib"; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $prog --cpu-seconds 0 --no-extra "$@" > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
echo "== $label"
python $GRAFT_REPO_ROOT/tools/kstats.py $O/trace/*/*kernel_stats.csv | grep -E "hg_(stream|verify|confirm|fin_|tile)" | head -12
rm -rf $O/trace
