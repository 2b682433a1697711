#!/usr/bin/env python3
"""GPU case runner executed in a CHILD process by tests/test_gpu_parity.py (a GPU memory fault ends the process that
caused it; the test then fails with the child's output instead of taking the whole test session down).

    python tests/gpu_cases.py guarded        parity cases on guarded text buffers (any over-read = GPU fault)
    python tests/gpu_cases.py many-sets      >= 48 distinct pattern sets through hyperscan(), HYPERGREP_POOL from the env
    python tests/gpu_cases.py huge           expressions beyond 1024 automaton positions (tests/huge_cases.py): buffer API on guarded
                                             buffers, the file API, block mode (hs_scan), mixed with ordinary expressions

Prints one JSON line; exit code 0 = every case matched the oracle.
"""
from __future__ import annotations

import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_py  # noqa: E402
import regex_gen  # noqa: E402


def oracle_hits(data, patterns, flags=None, ids=None, buffer_size=262140):
    rc, hits, nlines = oracle_py.scan_buffer(data, patterns, flags=flags, ids=ids, buffer_size=buffer_size)
    assert rc == 0
    return sorted(hits), nlines


def guarded_cases():
    """(name, text, patterns, flags, ids, buffer_size): what a text buffer's last, partial tile can meet."""
    rng = random.Random(20261004)
    base = regex_gen.random_text(rng, 1200, maxlen=200)
    # the round-1 fault (tools/fuzz_gpu.py seed 5002918): byte-aligned windows, a required literal of spaces (zero bytes past
    # the text fold onto ' '), one always-on pattern, one all-matches pattern, small scan buffer, partial last tile
    fault = (["yxy", "yx\\W=* {3}", "1(?:\\S1+\\-)0_."], [15, 14, 6], [1, 2, 2])
    for cut in (len(base), 121974 if len(base) >= 121974 else len(base) - 7, 16384 * 3, 16384 * 3 + 1, 16384 * 2 + 16, 16384 - 1, 4097, 100, 15):
        for bs in (1000, 262140):
            yield (f"fault-{cut}-{bs}", base[:cut], *fault, bs)
    spaces = regex_gen.random_text(rng, 600, maxlen=120).replace(b"_", b"    ")
    sets = [
        ("spaces-dword", ["        [a-c]", "(?i)a        b", "x    y    z"], None, [0, 1, 2]),           # >= 7-byte literals of spaces: dword windows
        ("zero-window", ["\\x00\\x00\\x00\\x00abcdefgh", "abc    xyz"], [14, 15], [0, 1]),                # a window value of zero
        ("short-dense", ["foo", "   ", "(?i)ab c"], [14, 14, 14], [0, 1, 2]),                              # byte-aligned probing, 3-byte windows
        ("every-second", ["abcab", "     x", "(?i)yz019"], None, [0, 1, 2]),                               # windows at every second byte
        ("always-on", [" +[a-c]", "[0-9]+\\.[0-9]+", "\\b[xyz]{2}\\b"], [14, 6, 14], [0, 1, 2]),           # no usable literal
    ]
    many = sorted({"".join(rng.choice("abcxyz019_-= ") for _ in range(rng.randint(8, 14))) for _ in range(700)})
    sets.append(("many-literals", [regex_gen_escape(w) for w in many], None, None))                        # filter beyond 16 KiB: the drain re-reads the text
    wide = sorted({"".join(rng.choice("0123456789abcdef ") for _ in range(12)) for _ in range(4200)})
    sets.append(("wide-filter", [regex_gen_escape(w) for w in wide], None, None))
    for name, pats, flags, ids in sets:
        for text in (spaces, base[:50000] + b"   ", spaces[:16384 * 2 + 5], b"abc    xyz", b" " * 40):
            for bs in (262140, 64):
                if bs == 64 and len(pats) > 100:
                    continue
                yield (f"{name}-{len(text)}-{bs}", text, pats, flags, ids, bs)


def regex_gen_escape(word: str) -> str:
    return "".join("\\" + c if c in ".-=" else c for c in word)


def run_guarded() -> dict:
    from hypergrep_amd import device

    n = 0
    bad = []
    compiled = {}
    arena = device.GuardedArena(1 << 20)
    for name, text, pats, flags, ids, bs in guarded_cases():
        want, nlines = oracle_hits(text, pats, flags, ids, buffer_size=bs)
        key = (tuple(pats), tuple(flags or ()), tuple(ids or ()))
        if key not in compiled:
            db = device.Database(pats, flags=flags, ids=ids)
            compiled[key] = (db, device.Scanner(db, 0))
        db, sc = compiled[key]
        stats = sc.scan(arena.place(text), len(text), buffer_size=bs)
        got = sorted(sc.hits())
        tail = arena.tail(len(text))
        if got != want or stats.n_lines != nlines:
            bad.append({"case": name, "got": len(got), "want": len(want), "lines": [stats.n_lines, nlines], "bytes_after_text": tail.hex(),
                        "extra": sorted(set(got) - set(want))[:6], "missing": sorted(set(want) - set(got))[:6]})
        n += 1
    return {"ok": not bad, "cases": n, "failed": bad[:8]}


def run_many_sets() -> dict:
    import hypergrep_amd
    from hypergrep_amd import device

    rng = random.Random(99)
    nsets = 48
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        for i in range(nsets):
            pairs = [regex_gen.anchored_pattern(rng) for _ in range(rng.randint(1, 6))]
            pats = [p for p, _ in pairs] + [regex_gen.random_pattern(rng) for _ in range(rng.randint(0, 2))]
            pats = [p for p in pats if oracle_py.check_patterns([p]) == 0] or ["needle_in_haystack"]
            ids = list(range(len(pats)))
            data = regex_gen.anchored_text(rng, [s for _, s in pairs], rng.choice([200, 2000])) + regex_gen.random_text(rng, 200, maxlen=100)
            path = os.path.join(tmp, f"f{i}.log")
            with open(path, "wb") as f:
                f.write(data)
            bs = rng.choice([262140, 1000, 64])
            count = rng.choice([1, 16, 500])
            os.environ["HYPERGREP_CHUNK_MB"] = rng.choice(["1", "256"])
            want_rc, want_rows, want_batches = oracle_py.scan_file(path, pats, None, ids, buffer_size=bs, buffer_count=count)
            rows, batches = [], []

            def on_match(matches, n, rows=rows, batches=batches):
                batches.append(n)
                for k in range(n):
                    rows.append((matches[k].line_number, matches[k].id, matches[k].line))

            rc = hypergrep_amd.scan(path, pats, on_match, ids=ids, buffer_size=bs, buffer_count=count)
            if (rc, rows, batches) != (want_rc, want_rows, want_batches):
                return {"ok": False, "set": i, "rc": [rc, want_rc], "rows": [len(rows), len(want_rows)]}
    return {"ok": True, "sets": nsets, **device.faceb_stats()}


def run_huge() -> dict:
    """tests/huge_cases.py on the GPU: the wave-cooperative routine (hg_huge.hip) against the oracle — tier 0 (confirm mode
    4, one run per piece and expression), always-on, all-matches mode, small scan buffers, text ending at an unmapped page."""
    import ctypes

    import hypergrep_amd
    from huge_cases import ACCEPTED_HUGE, REJECTED_HUGE, SCAN_CASES, case_text
    from hypergrep_amd import device, utils

    bad, n = [], 0
    for pat in ACCEPTED_HUGE:
        if hypergrep_amd.check_compatibility([pat]) != 0:
            bad.append({"case": "accept", "pattern": pat})
    for pat in REJECTED_HUGE:
        if hypergrep_amd.check_compatibility([pat]) != 4:
            bad.append({"case": "reject", "pattern": pat})
    arena = device.GuardedArena(2 << 20)
    for ci, (pats, flags, ids, kind) in enumerate(SCAN_CASES):
        db = device.Database(pats, flags=flags, ids=ids)
        sc = device.Scanner(db, 0)
        for seed in ((1,) if kind == "a32767" else (1, 2)):
            text = case_text(kind, seed)
            for bs in ((262140,) if kind == "a32767" else (262140, 2500)):
                want, nlines = oracle_hits(text, pats, flags, ids, buffer_size=bs)
                stats = sc.scan(arena.place(text), len(text), buffer_size=bs)
                got = sorted(sc.hits())
                if got != want or stats.n_lines != nlines:
                    bad.append({"case": f"buffer-{ci}-{seed}-{bs}", "patterns": pats, "got": len(got), "want": len(want), "lines": [stats.n_lines, nlines],
                                "extra": sorted(set(got) - set(want))[:6], "missing": sorted(set(want) - set(got))[:6]})
                n += 1
    # the largest automaton the compiler takes (393 204 positions: 12 288 state words, 96 KiB of LDS for the two state copies) beside
    # a small huge one (staged tables, the register-resident routine), on text where the monster's state stays small
    monster = "a{32767}" * 12
    big = [monster, "foo.{0,3000}bar", "a{3}b"]
    text = case_text("dotfoo", 7) + b"\n" + b"a" * 5000 + b"b\n" + b"aaab foo" + b"a" * 100 + b"bar\n"
    want, nlines = oracle_hits(text, big, None, [0, 1, 2])
    db = device.Database(big, ids=[0, 1, 2])
    sc = device.Scanner(db, 0)
    stats = sc.scan(arena.place(text), len(text))
    if sorted(sc.hits()) != want or stats.n_lines != nlines:
        bad.append({"case": "monster", "got": stats.n_hits, "want": len(want)})
    n += 1
    # a huge expression among ordinary ones of every confirm mode, shared ids; then through the file API (batches, line bytes)
    mixed = ["foo.{0,3000}bar", "needle_in_haystack", "fo+bar[0-9]", "\\bq[a-z]{1100,1300}\\b", "^[a-z]+ [a-z0-9 =._-]+$", "[a-z]{2000}x", "thread"]
    mflags = [14, 14, 14, 14, 14, 6, 14]
    mids = [0, 1, 0, 2, 3, 4, 1]
    text = case_text("dotfoo", 5) + b"\n" + case_text("context", 5) + b"needle_in_haystack foobar7\n" + case_text("az2000x", 5)
    want, nlines = oracle_hits(text, mixed, mflags, mids)
    db = device.Database(mixed, flags=mflags, ids=mids)
    sc = device.Scanner(db, 0)
    stats = sc.scan(arena.place(text), len(text))
    if sorted(sc.hits()) != want or stats.n_lines != nlines:
        bad.append({"case": "mixed-buffer", "got": stats.n_hits, "want": len(want)})
    n += 1
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        path = os.path.join(tmp, "huge.log")
        with open(path, "wb") as f:
            f.write(text)
        for bs, count in ((262140, 16), (3000, 5)):
            want_rc, want_rows, want_batches = oracle_py.scan_file(path, mixed, mflags, mids, buffer_size=bs, buffer_count=count)
            rows, batches = [], []

            def on_match(matches, k, rows=rows, batches=batches):
                batches.append(k)
                for j in range(k):
                    rows.append((matches[j].line_number, matches[j].id, matches[j].line))

            rc = hypergrep_amd.scan(path, mixed, on_match, flags=mflags, ids=mids, buffer_size=bs, buffer_count=count)
            if (rc, rows, batches) != (want_rc, want_rows, want_batches):
                bad.append({"case": f"file-{bs}", "rc": [rc, want_rc], "rows": [len(rows), len(want_rows)]})
            n += 1
    # block mode (Face A): hs_scan on whole blocks, newlines are ordinary bytes
    product = utils._get_hyperscanner_lib()  # pylint: disable=protected-access
    oracle = ctypes.CDLL(os.path.join(oracle_py.ORACLE_DIR, "_build", "libhs.so.5"))
    bpats = ["foo.{0,3000}bar", "[a-z]{2000}x", "(abc|def){200}", "plain_literal_here", "a.c"]
    bflags, bids = [14, 6, 14, 14, 6], [0, 1, 2, 3, 4]
    rng = random.Random(5)
    word = lambda k: "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(k)).encode()  # noqa: E731
    blocks = [b"foo" + word(2500) + b"bar plain_literal_here abc\n", b"fo" + word(100) + b"bar\n", word(2105) + b"x" + word(50) + b"x", b"abcdef" * 130 + b"!",
              b"foo\n" + word(10) + b"\nbar", word(1999) + b"x", b"x" * 9000 + b"foo" + b"y" * 3001 + b"bar"]
    import test_gpu_parity as tg  # (the libhs driver used by the Face A tests)

    got_ev, want_ev = tg._hs_events(product, bpats, bflags, bids, blocks), tg._hs_events(oracle, bpats, bflags, bids, blocks)  # pylint: disable=protected-access
    for k, (g, w) in enumerate(zip(got_ev, want_ev)):
        if g != w:
            bad.append({"case": f"block-{k}", "got": len(g), "want": len(w), "extra": sorted(set(g) - set(w))[:4], "missing": sorted(set(w) - set(g))[:4]})
        n += 1
    return {"ok": not bad, "cases": n, "block_events": sum(len(w) for w in want_ev), "failed": bad[:8]}


if __name__ == "__main__":
    res = {"guarded": run_guarded, "many-sets": run_many_sets, "huge": run_huge}[sys.argv[1]]()
    print(json.dumps(res), flush=True)
    sys.exit(0 if res.get("ok") else 1)
