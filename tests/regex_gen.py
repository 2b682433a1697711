"""Seeded random regex / text generators shared by the oracle and product parity tests.

Patterns use only constructs whose semantics are identical in PCRE (Hyperscan's dialect) and Python's
`re` on bytes, so Python `re` can serve as an independent existence-of-match check per line.
"""
from __future__ import annotations

import random

ALPHABET = "abcxyz019 _-=."
LITERALS = "abcxyz019_-= "


def _atom(rng: random.Random, depth: int) -> str:
    r = rng.random()
    if r < 0.45:
        c = rng.choice(LITERALS)
        return "\\" + c if c in ".-" else c
    if r < 0.55:
        return rng.choice(["[a-c]", "[xyz]", "[0-9]", "[^a]", "[^\\n]", "[a-c0-1_]", "[^0-9a-c]", "[-=_]"])
    if r < 0.65:
        return rng.choice(["\\d", "\\w", "\\s", "\\D", "\\W", "\\S"])
    if r < 0.70:
        return "."
    if r < 0.80 and depth < 2:
        n = rng.randint(1, 3)
        return "(" + "|".join(_seq(rng, depth + 1, rng.randint(1, 3)) for _ in range(n)) + ")"
    if r < 0.85 and depth < 2:
        return "(?:" + _seq(rng, depth + 1, rng.randint(1, 3)) + ")"
    c = rng.choice(LITERALS)
    return c


def _quant(rng: random.Random) -> str:
    r = rng.random()
    if r < 0.70:
        return ""
    if r < 0.78:
        return "*"
    if r < 0.86:
        return "+"
    if r < 0.92:
        return "?"
    a = rng.randint(0, 3)
    k = rng.random()
    if k < 0.4:
        return "{%d}" % max(a, 1)
    if k < 0.7:
        return "{%d,%d}" % (a, a + rng.randint(0, 3))
    return "{%d,}" % a


def _seq(rng: random.Random, depth: int, n: int) -> str:
    out = []
    for _ in range(n):
        if rng.random() < 0.08:
            out.append(rng.choice(["\\b", "\\B", "^", "$"]))
            continue
        out.append(_atom(rng, depth) + _quant(rng))
    return "".join(out)


def random_pattern(rng: random.Random) -> str:
    p = _seq(rng, 0, rng.randint(1, 6))
    if rng.random() < 0.1:
        p = "^" + p
    if rng.random() < 0.1:
        p = p + "$"
    return p


def random_line(rng: random.Random, maxlen: int = 24) -> bytes:
    n = rng.randint(0, maxlen)
    return "".join(rng.choice(ALPHABET) for _ in range(n)).encode()


def random_text(rng: random.Random, nlines: int, maxlen: int = 24, final_newline: bool = True) -> bytes:
    lines = [random_line(rng, maxlen) for _ in range(nlines)]
    data = b"\n".join(lines)
    if final_newline and nlines:
        data += b"\n"
    return data
