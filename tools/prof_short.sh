#!/bin/bash
# usage: tools/prof_short.sh <name> <GiB> <set index>   (on the GPU box; kernel stats of tools/short_bench.py -> gpurun_out/<name>_stats.txt)
name=$1
O=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/short_bench.py $2 $3 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/kstats.py $O/trace/*/*kernel_stats.csv > $GRAFT_REPO_ROOT/gpurun_out/${name}_stats.txt
rm -rf $O/trace
head -22 $GRAFT_REPO_ROOT/gpurun_out/${name}_stats.txt
