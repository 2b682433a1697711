#!/bin/bash
# usage: tools/ao_ab.sh <gib> <lib-or-""> set...: the always-on sets of tools/short_bench.py, one line per set
gib=$1; lib=$2; shift 2
for s in "$@"; do
  if [ -n "$lib" ]; then HG_LIB=$lib python tools/short_bench.py $gib $s 2>&1 | tail -1 | sed "s|^|set $s [$lib] |"; else python tools/short_bench.py $gib $s 2>&1 | tail -1 | sed "s|^|set $s [new] |"; fi
done
