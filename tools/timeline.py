#!/usr/bin/env python3
"""Print the kernel timeline of the LAST bench step from a rocprofv3 kernel_trace.csv (start/end in us relative to the
step's first kernel).  usage: timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("hg_stream", "hg_verify", "hg_confirm_fast", "hg_confirm_literal", "hg_confirm_generic", "hg_confirm_huge", "hg_tile_reduce", "hg_tile_spine", "hg_tile_apply", "hg_always", "hg_key", "hg_line_key",
              "hg_gather", "hg_keep", "radix", "select", "fillBuffer", "copyBuffer", "hg_synth", "onesweep", "histogram", "scan"):
        if k in n:
            return k
    return n[:40]
# the last step starts at the 4th-from-last... find last 'fillBuffer' preceding a run of stream kernels: use last N kernels after the final gap > 200us
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
# find index of the first kernel of the last step: walk back from the end while gaps are small
i = len(rows) - 1
while i > 0 and starts[i] - max(ends[:i][-20:]) < 300000:
    i -= 1
t0 = starts[i]
for r in rows[i:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:8.1f}  {short(r['Kernel_Name'])}")
