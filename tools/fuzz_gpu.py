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

def rich_pattern(rng):
    """Constructs beyond regex_gen's core set (both engines accept them): inline flags, POSIX classes, hex escapes, lazy and
    larger counted repeats, \\A \\z \\Z, \\Q...\\E."""
    parts = []
    for _ in range(rng.randint(1, 5)):
        r = rng.random()
        if r < 0.25:
            parts.append(regex_gen.random_pattern(rng))
        elif r < 0.35:
            parts.append("(?i:" + "".join(rng.choice("abcxyz") for _ in range(rng.randint(1, 4))) + ")")
        elif r < 0.45:
            parts.append(rng.choice(["[[:alpha:]]", "[[:digit:]]", "[[:space:]]", "[[:alnum:]_]", "[[:punct:]]"]) + rng.choice(["", "+", "*", "?", "{2}"]))
        elif r < 0.55:
            parts.append(rng.choice(["\\x61", "\\x{62}", "\\x30", "\\t", "\\x20"]))
        elif r < 0.65:
            parts.append(rng.choice("abcxyz01_") + rng.choice(["+?", "*?", "??", "{2,5}?", "{3,9}", "{6,}"]))
        elif r < 0.72:
            parts.append(rng.choice(["\\A", "\\z", "\\Z", "\\b", "\\B"]))
        elif r < 0.80:
            parts.append("\\Q" + "".join(rng.choice("ab.=-_ 0") for _ in range(rng.randint(1, 5))) + "\\E")
        elif r < 0.90:
            parts.append("(?:" + "|".join("".join(rng.choice("abcxyz019_-= ") for _ in range(rng.randint(1, 7))).replace("-", "\\-") for _ in range(rng.randint(2, 4))) + ")")
        else:
            parts.append("".join(rng.choice("abcxyz019_= ") for _ in range(rng.randint(3, 12))))
    return "".join(parts)


def huge_pattern(rng):
    """Expressions beyond 1024 automaton positions (sparse tables, the wave-cooperative routine): bounded repeats over the
    alphabet the random texts are made of, anchored or not by a literal, sometimes with boundary conditions."""
    n = rng.choice([1030, 1100, 1500, 2100, 4000])
    cls = rng.choice(["[a-c]", "[abcx01 ._-]", "[^y\\n]", ".", "[a-z0-9]", "(?:ab|c)"])
    head = rng.choice(["", "x", "ab", "xyz_", "\\b", "^", "0="])
    tail = rng.choice(["z", "y0", "_-", "$", "\\b", "1", " "])
    lo = rng.choice([0, 1, n // 2, n])
    return f"{head}{cls}{{{lo},{n}}}{tail}" if lo != n else f"{head}{cls}{{{n}}}{tail}"


def regex_gen_escape(word):
    return "".join("\\" + c if c in ".-=" else c for c in word)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
t0 = time.time()
cases = fails = 0
last_print = t0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    seed += 1
    if os.environ.get("HG_FUZZ_TRACE"):  # last line of the file = the case that was running when the process died
        with open(os.environ["HG_FUZZ_TRACE"], "a") as trace:
            trace.write(f"{seed - 1}\n")
    kind = rng.choice(["random", "random", "anchored", "mixed", "keywords", "keywords", "rich", "rich"])
    with_huge = rng.random() < 0.06
    if kind == "keywords":  # word lists: byte-aligned probing, 3-byte windows, short and long literals side by side
        lo, hi = rng.choice([(3, 3), (3, 5), (4, 6), (3, 9), (5, 12)])
        n = rng.choice([2, 8, 40, 300])
        pats = sorted({"".join(rng.choice(rng.choice(["abc", "abcxyz01", "abcxyz019_-= "])) for _ in range(rng.randint(lo, hi))) for _ in range(n)})
        pats = [regex_gen_escape(w) for w in pats]
        if rng.random() < 0.5:
            pats += [regex_gen.random_pattern(rng) for _ in range(rng.randint(1, 2))]
        samplers = []
    elif kind == "rich":
        pats = [rich_pattern(rng) for _ in range(rng.randint(1, 5))]
        samplers = []
    elif kind == "random":
        k = rng.randint(1, 6)
        pats = [regex_gen.random_pattern(rng) for _ in range(k)]
        samplers = []
    else:
        pairs = [regex_gen.anchored_pattern(rng) for _ in range(rng.randint(1, 12))]
        pats = [p for p, _ in pairs]
        samplers = [s for _, s in pairs]
        if kind == "mixed":
            pats += [regex_gen.random_pattern(rng) for _ in range(rng.randint(1, 3))]
    if with_huge:
        pats = pats[:4] + [huge_pattern(rng) for _ in range(rng.randint(1, 2))]
        samplers = samplers[:4]
    flags = [rng.choice([14, 14, 15, 10, 6, 12, 7, 2]) for _ in pats]
    ids = [rng.randint(0, 3) for _ in pats] if rng.random() < 0.7 else list(range(len(pats)))
    if oracle_py.check_patterns(pats, flags=flags) != 0:
        continue
    if samplers:
        data = regex_gen.anchored_text(rng, samplers, rng.choice([300, 3000]))
        if kind == "mixed":
            data += regex_gen.random_text(rng, 300, final_newline=rng.random() < 0.8)
    else:
        data = regex_gen.random_text(rng, rng.choice([40, 400, 3000]), maxlen=rng.choice([24, 24, 200]), final_newline=rng.random() < 0.8)
    if rng.random() < 0.3 or with_huge:  # NULs and a very long line
        b = bytearray(data)
        for _ in range(rng.randint(1, 8)):
            if b:
                b[rng.randrange(len(b))] = 0
        if rng.random() < 0.5 or with_huge:
            at = rng.randrange(len(b) + 1)
            b[at:at] = bytes(rng.choice(b"abcx01 ._-") for _ in range(rng.choice([5000, 20000, 40000])))
        data = bytes(b)
    bs = rng.choice([262140, 262140, 8, 64, 1000, 4096, 20000])
    only = os.environ.get("HG_FUZZ_ONLY", "")  # debugging aid: "faceb" / "dev" runs only that kind of case (same seeds)
    through_files = rng.random() < 0.25
    if (only == "faceb" and not through_files) or (only == "dev" and through_files):
        continue
    if through_files:  # the same case through the file API (Face B): batches, match limit, gzip, several ingest chunks
        import gzip
        import tempfile

        import hypergrep_amd

        big = data * rng.choice([1, 1, 3, 12]) if len(data) < 200000 else data
        count = rng.choice([1, 3, 16, 500])
        limit = rng.choice([0, 0, 1, 7, 100])
        fbs = rng.choice([262140, 262140, 64, 1000]) if b"\0" not in big else 262140
        with tempfile.NamedTemporaryFile(suffix=rng.choice([".log", ".gz"]), dir="/dev/shm", delete=False) as f:
            path = f.name
            f.write(gzip.compress(big, 1) if path.endswith(".gz") else big)
        os.environ["HYPERGREP_CHUNK_MB"] = rng.choice(["1", "1", "256"])
        try:
            want_rc, want_rows, want_batches = oracle_py.scan_file(path, pats, flags, ids, buffer_size=fbs, buffer_count=count, max_match_count=limit)
            rows, batches = [], []

            def on_match(matches, n, rows=rows, batches=batches):
                batches.append(n)
                for i in range(n):
                    rows.append((matches[i].line_number, matches[i].id, matches[i].line))

            rc = hypergrep_amd.scan(path, pats, on_match, flags=flags, ids=ids, buffer_size=fbs, buffer_count=count, max_match_count=limit)
        finally:
            os.unlink(path)
        cases += 1
        if (rc, rows, batches) != (want_rc, want_rows, want_batches):
            fails += 1
            first = next((i for i, (x, y) in enumerate(zip(rows, want_rows)) if x != y), min(len(rows), len(want_rows)))
            print(f"FACE-B MISMATCH seed {seed - 1} bytes={len(big)} bs={fbs} count={count} limit={limit} chunk_mb={os.environ['HYPERGREP_CHUNK_MB']} "
                  f"gz={path.endswith('.gz')} rc {rc}/{want_rc} rows {len(rows)}/{len(want_rows)} first diff at {first}: "
                  f"{rows[first:first + 2]} vs {want_rows[first:first + 2]} pats={pats} flags={flags} ids={ids}", flush=True)
        continue
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
