#!/bin/bash
# Per-call latency of hs_scan through the Face A artefact (hypergrep_amd/lib/libhs.so.5), driven with the reference shim's
# call sequence (tests/native/hs_call_order.c): a file of N lines = N hs_scan calls.  Runs on the GPU box.
#   tools/face_a_latency.sh  -> gpurun_out/face_a_latency.txt
R=$GRAFT_REPO_ROOT
cd $R
gcc -O2 -o tests/native/hs_call_order tests/native/hs_call_order.c -ldl || exit 1
python - <<'PY' > /dev/null
import sys
sys.path.insert(0, "tests")
import random, regex_gen
rng = random.Random(1)
open("/dev/shm/face_a_lines.txt", "wb").write(regex_gen.random_text(rng, 3000, maxlen=200) + b"needle_in_haystack here\n")
PY
{
  echo "# hs_scan latency through hypergrep_amd/lib/libhs.so.5 (one call per line, 3001 lines of <= 200 bytes; shim call order)"
  for pats in "14 0 needle_in_haystack" "14 0 needle_in_haystack 14 1 fo+bar[0-9]* 6 2 a.c" ; do
    echo "patterns: $pats"
    tests/native/hs_call_order hypergrep_amd/lib/libhs.so.5 /dev/shm/face_a_lines.txt 262140 --time -- $pats 2>&1 >/dev/null | tail -1
    echo "  (the oracle's CPU libhs, same calls:)"
    tests/native/hs_call_order oracle/_build/libhs.so.5 /dev/shm/face_a_lines.txt 262140 --time -- $pats 2>&1 >/dev/null | tail -1
  done
} > gpurun_out/face_a_latency.txt 2>&1
cat gpurun_out/face_a_latency.txt
