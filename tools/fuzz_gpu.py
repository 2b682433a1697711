#!/usr/bin/env python3
"""Extended randomized parity run on the GPU (not part of the test suite): random expressions, flags, ids, texts and
scan-buffer sizes against the oracle.     python tools/fuzz_gpu.py [seconds] [first seed]"""
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import oracle_py  # noqa: E402
import regex_gen  # noqa: E402
from test_gpu_parity import gpu_scan_buffer, oracle_hits  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
t0 = time.time()
cases = fails = 0
last_print = t0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    seed += 1
    kind = rng.choice(["random", "random", "anchored", "mixed"])
    if kind == "random":
        k = rng.randint(1, 6)
        pats = [regex_gen.random_pattern(rng) for _ in range(k)]
        samplers = []
    else:
        pairs = [regex_gen.anchored_pattern(rng) for _ in range(rng.randint(1, 12))]
        pats = [p for p, _ in pairs]
        samplers = [s for _, s in pairs]
        if kind == "mixed":
            pats += [regex_gen.random_pattern(rng) for _ in range(rng.randint(1, 3))]
    flags = [rng.choice([14, 14, 15, 10, 6, 12, 7, 2]) for _ in pats]
    ids = [rng.randint(0, 3) for _ in pats]
    if oracle_py.check_patterns(pats, flags=flags) != 0:
        continue
    if samplers:
        data = regex_gen.anchored_text(rng, samplers, rng.choice([300, 3000]))
        if kind == "mixed":
            data += regex_gen.random_text(rng, 300, final_newline=rng.random() < 0.8)
    else:
        data = regex_gen.random_text(rng, rng.choice([40, 400, 3000]), maxlen=rng.choice([24, 24, 200]), final_newline=rng.random() < 0.8)
    if rng.random() < 0.3:  # NULs and a very long line
        b = bytearray(data)
        for _ in range(rng.randint(1, 8)):
            if b:
                b[rng.randrange(len(b))] = 0
        if rng.random() < 0.5:
            at = rng.randrange(len(b) + 1)
            b[at:at] = bytes(rng.choice(b"abcx01 ._-") for _ in range(rng.choice([5000, 20000, 40000])))
        data = bytes(b)
    bs = rng.choice([262140, 262140, 8, 64, 1000, 4096, 20000])
    try:
        want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
        got, stats = gpu_scan_buffer(torch, data, pats, flags, ids, buffer_size=bs)
    except Exception as e:  # compile differences between the two engines are reported, not fatal
        print(f"seed {seed - 1}: {type(e).__name__}: {str(e)[:200]}  pats={pats} flags={flags}", flush=True)
        fails += 1
        continue
    cases += 1
    if got != want or stats.n_lines != nlines:
        fails += 1
        sg, sw = set(got), set(want)
        print(f"MISMATCH seed {seed - 1} kind={kind} bs={bs} pats={pats} flags={flags} ids={ids} lines {stats.n_lines}/{nlines} "
              f"missing={sorted(sw - sg)[:4]} extra={sorted(sg - sw)[:4]}", flush=True)
    if time.time() - last_print > 30:
        last_print = time.time()
        print(f"... {cases} cases, {fails} failures, seed {seed}", flush=True)
print(f"done: {cases} cases, {fails} failures, seeds up to {seed}")
