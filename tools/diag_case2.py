#!/usr/bin/env python3
import os, sys, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import torch
import gpu_cases, regex_gen
from hypergrep_amd import device

rng = random.Random(20261004)
base = regex_gen.random_text(rng, 1200, maxlen=200)
pats, flags, ids = ["yxy", "yx\\W=* {3}", "1(?:\\S1+\\-)0_."], [15, 14, 6], [1, 2, 2]
seq = [(n, bs) for n in (len(base), len(base) - 7, 16384 * 3, 16384 * 3 + 1, 16384 * 2 + 16, 16383, 4097, 100, 15) for bs in (1000, 262140)]

def one(sc, text, bs, guarded):
    want, nl = gpu_cases.oracle_hits(text, pats, flags, ids, bs)
    if guarded:
        buf = device.GuardedBuffer(text); ptr = buf.ptr
    else:
        t = torch.zeros(len(text) + 64, dtype=torch.uint8, device="cuda:0")
        t[:len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda(); torch.cuda.synchronize(); ptr = t.data_ptr()
    st = sc.scan(ptr, len(text), buffer_size=bs); got = sorted(sc.hits())
    ok = got == want and st.n_lines == nl
    extra = sorted(set(got) - set(want))[:8]
    print(f"  n={len(text)} bs={bs} ok={ok} lines {st.n_lines}/{nl} hits {len(got)}/{len(want)} cands={st.n_candidates} raw={st.n_raw_hits} reruns={st.reruns} extra={extra}", flush=True)
    if guarded: buf.free()

for mode in ("reuse+guarded", "reuse+torch", "fresh+guarded"):
    print(mode, flush=True)
    db = device.Database(pats, flags=flags, ids=ids); sc = device.Scanner(db, 0)
    for n, bs in seq:
        if mode.startswith("fresh"):
            db = device.Database(pats, flags=flags, ids=ids); sc = device.Scanner(db, 0)
        one(sc, base[:n], bs, "guarded" in mode)
