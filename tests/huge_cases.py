"""Expressions of more than 1024 automaton positions ("huge": sparse tables, hg_db.h HgHugeHeader) and texts that exercise
them: hits, near misses, lines longer than the repeat, NULs, CRLF.  Shared by the host replay (test_huge_patterns.py) and
the GPU parity tests (test_gpu_parity.py).  Reference call sites: hs_compile_multi hypergrep/lib/c/hyperscanner.c:136,
check_patterns :154-167; Hyperscan's bounded-repeat limit is 32767."""
from __future__ import annotations

import random

# these gave rc 4 in the product and rc 0 in the oracle
ACCEPTED_HUGE = [
    "[a-z]{2000}x", ".{0,3000}foo", "foo.{0,3000}bar", "(abc|def){200}", "a{32767}",
    "(a?){3000}b", "(a*){2,3000}b", "x(ab?c?){1500}y", "[a-z]{32767}[0-9]{32767}", "\\bq[a-z]{1100,1300}\\b", "^.{1500}$",
    "(?i)head[a-z ]{1,2000}tail", "(foo|ba[rz]){2,400}!", "k(?:[0-9]{3}-){300}z", "needle[^\\n]{0,5000}thread",
]
REJECTED_HUGE = ["a{32768}", "a{1,32768}", "(abcdefgh|ijklmnop){32767}", "((ab){500}){500}"]


def case_text(kind: str, seed: int) -> bytes:
    """A text of ~200 lines for the expression family `kind`, with matching lines, near misses and long lines."""
    rng = random.Random(seed)
    az = "abcdefghijklmnopqrstuvwxyz"

    def word(n, alphabet=az):
        return "".join(rng.choice(alphabet) for _ in range(n))

    lines: list[bytes] = []
    for _ in range(60):
        lines.append((word(rng.randint(5, 40)) + " " + word(rng.randint(5, 60), az + "0123456789 =._-")).encode())
    if kind == "az2000x":  # [a-z]{2000}x
        for n in (1999, 2000, 2001, 2500, 4100):
            lines.append((word(n) + "x").encode())
            lines.append((word(n) + "y").encode())
            lines.append((word(n // 2) + "7" + word(n - n // 2) + "x").encode())
        lines.append(("x" * 6000).encode())
        lines.append((word(1990) + "X" + word(3000) + "x tail").encode())
    elif kind == "dotfoo":  # .{0,3000}foo / foo.{0,3000}bar / needle[^\n]{0,5000}thread
        for gap in (0, 1, 7, 2999, 3000, 3001, 3500, 5000, 5001, 7000):
            filler = word(gap, az + " :=")
            lines.append(("foo" + filler + "bar").encode())
            lines.append(("fo" + filler + "bar foo").encode())
            lines.append(("needle" + filler + "thread").encode())
            lines.append(("needle" + filler + "thre ad").encode())
            lines.append((filler + "foo").encode())
        lines.append(b"foo\x00" + word(10).encode() + b"bar")
        lines.append(b"\x00\x00foo" + word(20).encode() + b"bar\r")
        lines.append(("foo" + word(1000) + "foo" + word(2500) + "bar" + word(800) + "bar").encode())
    elif kind == "abcdef":  # (abc|def){200} and friends
        for n in (199, 200, 201, 400, 450):
            lines.append("".join(rng.choice(["abc", "def"]) for _ in range(n)).encode())
            lines.append(("".join(rng.choice(["abc", "def"]) for _ in range(n // 2)) + "abd" + "".join(rng.choice(["abc", "def"]) for _ in range(n - n // 2))).encode())
        for n in (1, 2, 3, 399, 400, 401, 500):
            lines.append(("".join(rng.choice(["foo", "bar", "baz"]) for _ in range(n)) + "!").encode())
            lines.append(("".join(rng.choice(["foo", "bar", "bax"]) for _ in range(n)) + "!").encode())
        for n in (299, 300, 301):
            lines.append(("k" + "".join("%03d-" % rng.randint(0, 999) for _ in range(n)) + "z").encode())
    elif kind == "a32767":  # a{32767}: needs lines longer than the default piece? no: pieces of 262139 bytes hold it
        lines = lines[:10]
        for n in (32766, 32767, 32769):
            lines.append(b"a" * n)
        lines.append(b"a" * 16000 + b"b" + b"a" * 16767)
        lines.append((word(32767) + word(32767, "0123456789")).encode())
        lines.append((word(32766) + word(32768, "0123456789")).encode())
    elif kind == "optional":  # (a?){3000}b, (a*){2,3000}b, x(ab?c?){1500}y
        for n in (0, 1, 2999, 3000, 3001, 4000):
            lines.append(b"a" * n + b"b")
            lines.append(b"c" + b"a" * n + b"b")
            lines.append(b"a" * n + b"c")
        for n in (1499, 1500, 1501):
            body = "".join("a" + ("b" if rng.random() < 0.5 else "") + ("c" if rng.random() < 0.5 else "") for _ in range(n))
            lines.append(("x" + body + "y").encode())
            lines.append(("x" + body + "by").encode())
    elif kind == "context":  # \bq[a-z]{1100,1300}\b, ^.{1500}$, (?i)head[a-z ]{1,2000}tail
        for n in (1099, 1100, 1200, 1300, 1301):
            lines.append(("pre q" + word(n) + " post").encode())
            lines.append(("preq" + word(n) + " post").encode())
            lines.append(("pre q" + word(n) + "_post").encode())
        for n in (1499, 1500, 1501):
            lines.append(word(n, az + " 0123").encode())
        for n in (0, 1, 500, 2000, 2001):
            lines.append(("HeAd" + word(n, az + " ") + "TAIL").encode())
            lines.append(("head" + word(n, az + " ") + "7tail").encode())
    rng.shuffle(lines)
    return b"\n".join(lines) + (b"\n" if seed % 2 else b"")


# (expression list, flags, ids, text kind)
SCAN_CASES = [
    (["[a-z]{2000}x"], None, None, "az2000x"),
    (["[a-z]{2000}x", "x[a-z]{3}"], [6, 14], [3, 1], "az2000x"),  # all-matches mode for the huge one
    ([".{0,3000}foo", "foo.{0,3000}bar", "needle[^\\n]{0,5000}thread"], None, [0, 1, 2], "dotfoo"),
    (["foo.{0,3000}bar"], [6], [5], "dotfoo"),
    (["(abc|def){200}", "(foo|ba[rz]){2,400}!", "k(?:[0-9]{3}-){300}z"], None, [0, 1, 2], "abcdef"),
    (["a{32767}", "[a-z]{32767}[0-9]{32767}"], None, [0, 1], "a32767"),
    (["(a?){3000}b", "(a*){2,3000}b", "x(ab?c?){1500}y"], None, [0, 1, 2], "optional"),
    (["(a?){3000}b"], [6], None, "optional"),
    (["\\bq[a-z]{1100,1300}\\b", "^.{1500}$", "(?i)head[a-z ]{1,2000}tail"], None, [0, 1, 2], "context"),
    (["^.{1500}$", "plainliteral", "q[a-z]+ post"], [6, 14, 14], [0, 1, 2], "context"),
]
