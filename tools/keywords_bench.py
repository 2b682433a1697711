#!/usr/bin/env python3
"""Keyword lists (many short literals) through the device API: which filter the compiler picks and what it costs.
    python tools/keywords_bench.py [GiB]"""
import random
import sys
import time

sys.path.insert(0, "/root/repo")
import torch

from hypergrep_amd import benchspec, device

_, needles, hpm = benchspec.c3_spec()
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nbytes = gib << 30
text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
device.synth_device(text.data_ptr(), nbytes, 77, needles, hpm)
torch.cuda.synchronize()
sample = bytes(text[: 4 << 20].cpu().numpy())
rng = random.Random(5)
syll = ["ba", "ne", "ko", "ri", "tu", "sa", "mo", "li", "fe", "du", "ga", "po", "xi", "ze", "wa", "cy", "qu", "th", "er", "on", "st", "in"]


def word(lo, hi):
    w = ""
    while len(w) < lo:
        w += rng.choice(syll)
    return w[: rng.randint(lo, hi)]


for n, lo, hi in ((100, 4, 8), (1000, 4, 8), (5000, 4, 10), (1000, 3, 5), (12000, 5, 10)):
    pats = sorted({word(lo, hi) for _ in range(n)}) + ["retry", "cache"]
    t0 = time.perf_counter()
    db = device.Database(pats, ids=list(range(len(pats))))
    db.tune(sample)
    t_compile = time.perf_counter() - t0
    info = db.info()
    sc = device.Scanner(db, 0)
    sc.scan(text.data_ptr(), nbytes)
    t = time.perf_counter()
    for _ in range(3):
        st = sc.scan(text.data_ptr(), nbytes)
    dt = (time.perf_counter() - t) / 3
    print(f"{len(pats):6d} words of {lo}-{hi} bytes: byte_windows={info['byte_windows']} always_on={info['n_always_on']} windows={info['n_windows']} "
          f"compile {t_compile:.2f} s | {gib / dt:8.1f} GiB/s  hits={st.n_hits} cands={st.n_candidates} stream={st.ms_stream:.3f} ms total={st.ms_total:.3f} ms", flush=True)
