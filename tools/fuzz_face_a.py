#!/usr/bin/env python3
"""Randomized parity of the libhs face (hs_compile_multi / hs_scan in block mode, GPU-backed) against the oracle's libhs:
random expressions, flags, ids and blocks (newlines and NULs are ordinary bytes in block mode).
    python tools/fuzz_face_a.py [seconds] [first seed]"""
import ctypes
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import torch  # noqa: E402,F401

import oracle_py  # noqa: E402
import regex_gen  # noqa: E402
from hypergrep_amd import utils  # noqa: E402
from test_gpu_parity import _hs_events  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
oracle_py.build()
product = utils._get_hyperscanner_lib()
oracle = ctypes.CDLL(os.path.join(oracle_py.ORACLE_DIR, "_build", "libhs.so.5"))
t0 = time.time()
cases = fails = 0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    seed += 1
    if rng.random() < 0.5:
        pats = [regex_gen.random_pattern(rng) for _ in range(rng.randint(1, 5))]
        samplers = []
    else:
        pairs = [regex_gen.anchored_pattern(rng) for _ in range(rng.randint(1, 6))]
        pats = [p for p, _ in pairs] + [regex_gen.random_pattern(rng) for _ in range(rng.randint(0, 2))]
        samplers = [s for _, s in pairs]
    flags = [rng.choice([14, 14, 15, 10, 6, 12, 7, 2]) for _ in pats]
    ids = [rng.randint(0, 3) for _ in pats]
    if oracle_py.check_patterns(pats, flags=flags) != 0:
        continue
    blocks = []
    for _ in range(rng.randint(1, 4)):
        if samplers and rng.random() < 0.6:
            b = regex_gen.anchored_text(rng, samplers, rng.choice([3, 40, 400]))
        else:
            b = regex_gen.random_text(rng, rng.choice([1, 5, 60, 600]), final_newline=rng.random() < 0.7)
        if rng.random() < 0.2 and b:
            bb = bytearray(b)
            bb[rng.randrange(len(bb))] = 0
            b = bytes(bb)
        blocks.append(b if b else b"x")
    try:
        got = _hs_events(product, pats, flags, ids, blocks)
        want = _hs_events(oracle, pats, flags, ids, blocks)
    except AssertionError as e:
        print(f"seed {seed - 1}: call failed ({e}) pats={pats} flags={flags}", flush=True)
        fails += 1
        continue
    cases += 1
    if [sorted(g) for g in got] != [sorted(w) for w in want]:
        fails += 1
        k = next(i for i in range(len(blocks)) if sorted(got[i]) != sorted(want[i]))
        print(f"MISMATCH seed {seed - 1} block {k} ({len(blocks[k])} bytes) pats={pats} flags={flags} ids={ids} "
              f"extra={sorted(set(got[k]) - set(want[k]))[:4]} missing={sorted(set(want[k]) - set(got[k]))[:4]}", flush=True)
    elif got != want:
        fails += 1
        print(f"ORDER seed {seed - 1} pats={pats} flags={flags} ids={ids}", flush=True)
print(f"done: {cases} cases, {fails} failures, seeds up to {seed}")
