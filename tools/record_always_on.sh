#!/bin/bash
# usage: tools/record_always_on.sh <out file under gpurun_out/> [baseline lib]   (GPU box)
# The always-on sets of tools/short_bench.py at 8 GiB: this build (and a baseline library for a same-box A/B), the kernel
# table of three sets and the counters of the automaton kernel.  The summary is copied into profiles/ by hand.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; base=$2
{
  echo "# tools/record_always_on.sh: always-on sets of tools/short_bench.py, 8 GiB of the benchmark's synthetic log, MI355X"
  echo "# set 4: [0-9]+\\.[0-9]+ (a hit on every line)   8: \\b[0-9]{3}\\b   9: four expressions, 160 M hits"
  echo "# set 10: [a-z]+@[a-z]+ (no hits)   11: four expressions, one with \\b, no hits"
  tools/ao_ab.sh 8 "" 4 8 9 10 11
  [ -n "$base" ] && tools/ao_ab.sh 8 $base 4 8 9 10 11
  for s in 9 10 11; do
    tools/prof_short.sh ao_rec_k$s 8 $s > /dev/null 2>&1
    echo; echo "## kernels, set $s (rocprofv3 --kernel-trace --stats; 1 warm-up + 3 timed passes, plus the text generator)"
    head -9 $GRAFT_REPO_ROOT/gpurun_out/ao_rec_k${s}_stats.txt
  done
  for s in 10 11; do
    tools/pmc_short.sh ao_rec_pmc$s 8 $s SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU > $GRAFT_REPO_ROOT/gpurun_out/ao_rec_pmc$s.txt 2>&1
    tools/pmc_short.sh ao_rec_l2$s 8 $s TCP_TCC_READ_REQ_sum > $GRAFT_REPO_ROOT/gpurun_out/ao_rec_l2$s.txt 2>&1
    echo; echo "## counters of hg_always_on_fast_kernel, set $s, per pass over 8 GiB (rocprofv3 --pmc, own passes)"
    grep -h always_on_fast $GRAFT_REPO_ROOT/gpurun_out/ao_rec_pmc$s.txt $GRAFT_REPO_ROOT/gpurun_out/ao_rec_l2$s.txt
  done
} > $out 2>&1
cat $out
