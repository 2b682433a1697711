#!/usr/bin/env python3
"""GPU case runner executed in a CHILD process by tests/test_gpu_parity.py (a GPU memory fault ends the process that
caused it; the test then fails with the child's output instead of taking the whole test session down).

    python tests/gpu_cases.py guarded        parity cases on guarded text buffers (any over-read = GPU fault)
    python tests/gpu_cases.py many-sets      >= 48 distinct pattern sets through hyperscan(), HYPERGREP_POOL from the env

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


if __name__ == "__main__":
    res = {"guarded": run_guarded, "many-sets": run_many_sets}[sys.argv[1]]()
    print(json.dumps(res), flush=True)
    sys.exit(0 if res.get("ok") else 1)
