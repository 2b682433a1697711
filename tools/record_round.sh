#!/bin/bash
# Runs on the GPU box: official bench line + rocprofv3 kernel stats + HBM traffic counters for the same command.
# Everything lands under gpurun_out/record/ ; copy what should be judged into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/record
mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --cpu-seconds 0 > $O/trace.log 2>&1 || exit 1
python $R/tools/kstats.py $O/trace/*/*kernel_stats.csv > $O/kernel_stats.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/pmc_write.log 2>&1 || exit 1
python - <<PY > $O/hbm_traffic.txt
import csv, glob, collections
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "pmc_fetch" if name == "FETCH_SIZE" else "pmc_write"
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"]
            k = "hg_stream_join_kernel" if "hg_stream_join" in k else "hg_stream_kernel" if "hg_stream" in k else ("hg_verify_kernel" if "hg_verify" in k else ("hg_confirm*" if "hg_confirm" in k else None))
            if not k: continue
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print(f"{name:11s} {k:20s} launches={n:3d} avg_per_launch_KB={v/n:.6g}")
    out[name] = agg
import json
f, w = out["FETCH_SIZE"].get("hg_stream_kernel"), out["WRITE_SIZE"].get("hg_stream_kernel")
# average duration of hg_stream_kernel in the --kernel-trace --stats run of the same command (bench.py prices its roofline fraction with it)
kms = None
for line in open("$O/kernel_stats.txt"):
    parts = line.split()
    if parts and parts[0] == "hg_stream_kernel":
        kms = round(float(parts[3]) / 1e3, 5)
if f and w:
    per_launch = int((2 * f[1] / f[0] + w[1] / w[0]) * 1024)
    json.dump({"workload": "c3", "gib": 32, "kernel": "hg_stream_kernel", "hbm_bytes_per_launch": per_launch,
               "kernel_ms_rocprof": kms,
               "kernel_ms_rocprof_source": "tools/record_round.sh: rocprofv3 --kernel-trace --stats -- python bench.py --cpu-seconds 0, average over the kernel's launches (the summary is committed under profiles/)",
               "source": "tools/record_round.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, averaged over the "
                         "kernel's launches; FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half, MI355X_MICROARCH.md HBM section)"},
              open("$O/hbm_traffic_latest.json", "w"), indent=1)
print("# FETCH_SIZE is in KB of 64-B requests; on gfx950 a wide coalesced streaming read counts HALF its bytes")
print("# (MI355X_MICROARCH.md, HBM section): HBM read bytes of hg_stream_kernel = 2 x FETCH_SIZE x 1024.")
PY
# the official line last: bench.py prices the roofline fraction with the rocprofv3 duration and the PMC traffic recorded just now
cp $O/hbm_traffic_latest.json $R/profiles/hbm_traffic_latest.json
cd $R
timeout -k 10 400 python bench.py 2>/dev/null | tail -1 > $O/bench.json || exit 1
cat $O/bench.json; head -12 $O/kernel_stats.txt; cat $O/hbm_traffic.txt
