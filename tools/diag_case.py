#!/usr/bin/env python3
"""Diagnose one guarded-buffer parity case: fresh scanners, torch buffer vs guarded buffer, pattern subsets."""
import os, sys, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import torch
import gpu_cases, regex_gen
from hypergrep_amd import device

rng = random.Random(20261004)
base = regex_gen.random_text(rng, 1200, maxlen=200)
pats, flags, ids = ["yxy", "yx\\W=* {3}", "1(?:\\S1+\\-)0_."], [15, 14, 6], [1, 2, 2]

def run(text, p, f, i, bs, guarded, tail=b""):
    want, nl = gpu_cases.oracle_hits(text, p, f, i, bs)
    db = device.Database(p, flags=f, ids=i); sc = device.Scanner(db, 0)
    if guarded:
        buf = device.GuardedBuffer(text); ptr = buf.ptr
    else:
        t = torch.zeros(len(text) + 64, dtype=torch.uint8, device="cuda:0")
        t[:len(text) + len(tail)] = torch.frombuffer(bytearray(text + tail), dtype=torch.uint8).cuda(); torch.cuda.synchronize(); ptr = t.data_ptr()
    st = sc.scan(ptr, len(text), buffer_size=bs); got = sorted(sc.hits())
    ok = got == want and st.n_lines == nl
    extra = sorted(set(got) - set(want))[:6]; missing = sorted(set(want) - set(got))[:6]
    print(f"n={len(text)} bs={bs} guarded={guarded} tail={len(tail)} pats={len(p)} ok={ok} lines {st.n_lines}/{nl} hits {len(got)}/{len(want)} extra={extra} missing={missing} info={db.info()}", flush=True)
    return ok

for n in (49153, 49152 + 16, 49152 + 17, 16385, 32769):
    for bs in (1000, 262140):
        run(base[:n], pats, flags, ids, bs, True)
        run(base[:n], pats, flags, ids, bs, False)
        run(base[:n], pats, flags, ids, bs, False, tail=b"\n" * 15)
for sub in ([0], [1], [2], [0, 1], [1, 2]):
    run(base[:49153], [pats[k] for k in sub], [flags[k] for k in sub], [ids[k] for k in sub], 1000, True)
