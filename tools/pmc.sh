#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <bench args>
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$out/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hg_stream" not in k and "hg_confirm" not in k and "hg_verify" not in k: continue
        if "generic" in k: k = k.replace("hg_confirm", "hg_cgeneric")
        k = ("stream" if "hg_stream" in k else ("verify" if "hg_verify" in k else ("ctx" if "ctx" in k else "confirm")), r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print(f"{k[0]:8s} {k[1]:28s} n={n:3d} avg={v/n:.4g}")
PY
