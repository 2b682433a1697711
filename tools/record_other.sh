#!/bin/bash
# Runs on the GPU box: the other configurations' summary lines + config 5's per-kernel table and timeline -> gpurun_out/record/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/record
mkdir -p $O
cd $R
{
  echo "# other configurations, one bench.py run each (tools/ab.sh line format), final round-3 build"
  bash tools/ab.sh "c3-32g||-|--steps 10 --no-extra" "c3-64g||-|--steps 5 --no-extra --gib 64" "c2-4g||-|--steps 10 --no-extra --workload c2 --gib 4" \
    "c2-32g||-|--steps 10 --no-extra --workload c2" "c1-32g||-|--steps 10 --no-extra --workload c1" "c5-32g||-|--steps 5 --no-extra --workload c5" \
    "c5-32g||-|--steps 5 --no-extra --workload c5" "c5-4g||-|--steps 10 --no-extra --workload c5 --gib 4"
  echo
  echo "# c5 (4096 literals, 32 GiB) per-kernel table, rocprofv3 --kernel-trace --stats (tools/kstat_ab.sh)"
  tools/kstat_ab.sh c5 - --steps 5 --workload c5 | grep -v "^=="
} > $O/other_configs.txt 2>&1
bash tools/trace.sh c5final -- --workload c5 --steps 3 --warmup 2 --no-extra > /dev/null 2>&1
{
  echo "# kernel timeline of one bench step (chunked pipeline, c5: 4096 literals, 32 GiB): start_us end_us dur_us kernel; rocprofv3 --kernel-trace via tools/trace.sh"
  awk 'BEGIN{p=0} /hg_reset_kernel/{p=1; n=0} p{buf[n++]=$0} END{for(i=0;i<n;i++)print buf[i]}' $R/gpurun_out/c5final_timeline.txt
} > $O/timeline_c5.txt
bash tools/trace.sh c3final -- --steps 3 --warmup 2 --no-extra > /dev/null 2>&1
{
  echo "# kernel timeline of one bench step (chunked pipeline, c3, 32 GiB): start_us end_us dur_us kernel; rocprofv3 --kernel-trace via tools/trace.sh"
  awk 'BEGIN{p=0} /hg_reset_kernel/{p=1; n=0} p{buf[n++]=$0} END{for(i=0;i<n;i++)print buf[i]}' $R/gpurun_out/c3final_timeline.txt
} > $O/timeline_c3.txt
cat $O/other_configs.txt
