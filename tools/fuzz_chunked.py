#!/usr/bin/env python3
"""Randomized property run on the GPU: the chunked two-stream pipeline and the segmented scan must deliver exactly the records of the single
pass — random pattern sets (dword-aligned, byte-aligned, always-on, mixed), sizes, chunk sizes, scan buffers, line bases.
    python tools/fuzz_chunked.py [seconds] [first seed]"""
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from hypergrep_amd import benchspec, device  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c3, needles, hpm = benchspec.c3_spec()
short = ["retry", "cache", "=77", "ok", "GET", "POST", "eu-west", "(?i)TiMeOuT", "shard=[0-9]+", "[0-9]{4}7", "=7[0-9]? ", "\\bdone\\b", "us-east$",
         "closed.*accepted", "^[A-Z]+ ", "x", "latency_ms=9"]
t0 = time.time()
cases = fails = 0
last = t0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    seed += 1
    kind = rng.choice(["c3", "c3+short", "short", "few"])
    if kind == "c3":
        pats = rng.sample(c3, rng.choice([8, 64, 256]))
    elif kind == "c3+short":
        pats = rng.sample(c3, rng.choice([8, 64])) + rng.sample(short, rng.randint(1, 4))
    elif kind == "short":
        pats = rng.sample(short, rng.randint(1, 6))
    else:
        pats = rng.sample(c3, 2) + rng.sample(short, 1)
    flags = [rng.choice([14, 14, 14, 6, 15]) for _ in pats]
    ids = list(range(len(pats))) if rng.random() < 0.5 else [rng.randint(0, 3) for _ in pats]
    nbytes = rng.randint(40 << 20, 300 << 20) + rng.randint(0, 99999)
    always_on_heavy = any(p in ("x", "ok", "^[A-Z]+ ", "[0-9]{4}7", "=7[0-9]? ") for p in pats)
    if always_on_heavy:
        nbytes = min(nbytes, 64 << 20)
    bs = rng.choice([262140, 262140, 262140, 4096, 100])
    line_base = rng.choice([0, 77, 1 << 33])
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, 9000 + seed, needles, hpm * rng.choice([1, 2, 8]))
    torch.cuda.synchronize()
    try:
        db = device.Database(pats, flags=flags, ids=ids)
    except Exception as e:
        print(f"seed {seed - 1}: compile: {e}", flush=True)
        continue
    ntiles = (nbytes + 16383) // 16384
    chunk_tiles = max(1024, ntiles // rng.choice([2, 3, 5, 9]) // 1024 * 1024)  # (the engine takes multiples of 1024 tiles)
    if ntiles <= chunk_tiles:
        continue

    def run():
        sc = device.Scanner(db, 0)  # (the engine's knobs are read when a scanner is created)
        st = sc.scan(text.data_ptr(), nbytes, buffer_size=bs, line_base=line_base)
        buf = torch.empty((max(st.n_hits, 1), 2), dtype=torch.int64, device="cuda:0")
        sc.copy_hits_to(buf.data_ptr(), st.n_hits)
        torch.cuda.synchronize()
        return st, buf[: st.n_hits]

    os.environ["HG_CHUNK_TILES"] = str(1 << 30)
    one, a = run()
    os.environ["HG_CHUNK_TILES"] = str(chunk_tiles)
    many, b = run()
    # ... and in segments (a pass may hold fewer reports / pipeline chunks than the text needs)
    seg_same = True
    if one.n_raw_hits >= 8192 and rng.random() < 0.7:
        how = rng.choice(["hits", "chunks", "both"])
        if how in ("hits", "both"):
            os.environ["HG_HIT_LIMIT"] = str(max(1024, one.n_raw_hits // rng.choice([2, 3, 7])))
        if how in ("chunks", "both"):
            os.environ["HG_MAX_CHUNKS"] = str(rng.choice([1, 2]))
        try:
            seg, c = run()
            seg_same = (one.n_hits, one.n_lines) == (seg.n_hits, seg.n_lines) and bool((a == c).all()) and seg.stream_launches >= 2
        except Exception as e:
            seg_same = False
            print(f"seed {seed - 1}: segmented scan: {e}", flush=True)
        os.environ.pop("HG_HIT_LIMIT", None)
        os.environ.pop("HG_MAX_CHUNKS", None)
        if not seg_same:
            fails += 1
            print(f"SEGMENT MISMATCH seed {seed - 1} kind={kind} how={how} bytes={nbytes} bs={bs} chunk_tiles={chunk_tiles} raw={one.n_raw_hits} pats={pats[:6]}", flush=True)
    cases += 1
    same = (one.n_hits, one.n_lines) == (many.n_hits, many.n_lines) and bool((a == b).all())
    if not same or many.stream_launches < 2:
        fails += 1
        print(f"MISMATCH seed {seed - 1} kind={kind} n={len(pats)} bytes={nbytes} bs={bs} chunk_tiles={chunk_tiles} launches={many.stream_launches} "
              f"hits {one.n_hits}/{many.n_hits} lines {one.n_lines}/{many.n_lines} info={db.info()} pats={pats[:6]}", flush=True)
    del text, db
    if time.time() - last > 30:
        last = time.time()
        print(f"... {cases} cases, {fails} failures, seed {seed}", flush=True)
print(f"done: {cases} cases, {fails} failures, seeds up to {seed}")
