#!/bin/bash
# usage: tools/quick_bench.sh <label> [bench args...]: one line per run: label value ms kernel_ms pipeline_frac cands hits
label=$1; shift
python bench.py --cpu-seconds 0 --no-extra "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$label', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['pipeline_frac'], d['pipeline']['candidates'], d['config']['hits'])"
