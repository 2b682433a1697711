#!/bin/bash
# usage (GPU box): tools/e2e_sweep.sh   — Face B end-to-end rate for a few reader-thread / chunk-size settings
cd $GRAFT_REPO_ROOT
for t in 8 16; do
  for c in 64 128 256; do
    echo "== threads=$t chunk_mb=$c"
    HYPERGREP_READ_THREADS=$t HYPERGREP_CHUNK_MB=$c timeout -k 10 200 python tools/e2e_bench.py --gib 4 --reps 3 2>&1 | grep "rep [12]"
  done
done
