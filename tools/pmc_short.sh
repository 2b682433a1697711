#!/bin/bash
# usage: tools/pmc_short.sh <outdir> <GiB> <set index> <counters...>   (GPU box; per-kernel averages of tools/short_bench.py)
out=$1; gib=$2; idx=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python $GRAFT_REPO_ROOT/tools/short_bench.py $gib $idx > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
python - <<PY
import csv, glob, collections, re
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$out/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        m = re.search(r"(hg_\w+)", r["Kernel_Name"])
        if not m or "synth" in m.group(1): continue
        k = (m.group(1), r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print(f"{k[0]:28s} {k[1]:26s} n={n:3d} avg={v/n:.5g}")
PY
