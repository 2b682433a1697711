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


# ---- patterns anchored by a long literal (the prefilter tier), with strings that match them and near-misses
_PARTS = [
    ("[0-9]{1,3}", lambda r: "".join(r.choice("0123456789") for _ in range(r.randint(1, 3)))),
    ("[a-f]+", lambda r: "".join(r.choice("abcdef") for _ in range(r.randint(1, 6)))),
    ("(?:foo|ba+r)", lambda r: r.choice(["foo", "bar", "baaar"])),
    ("x?", lambda r: r.choice(["", "x"])),
    (".{0,4}", lambda r: "".join(r.choice("qrs tuv") for _ in range(r.randint(0, 4)))),
    ("[A-Z_]{2,5}", lambda r: "".join(r.choice("ABCXYZ_") for _ in range(r.randint(2, 5)))),
    ("\\b", lambda r: ""),
    (" +", lambda r: " " * r.randint(1, 3)),
    ("(?:ab|cd){1,2}", lambda r: "".join(r.choice(["ab", "cd"]) for _ in range(r.randint(1, 2)))),
    ("[^,;]", lambda r: r.choice("abz09 _")),
]


def anchored_pattern(rng):
    """(pattern, sampler): a random expression around a literal of 8..16 bytes; sampler(rng) returns a string that
    usually matches (boundary parts may veto it) or, one time in three, a near-miss of it."""
    lit = "".join(rng.choice("ghijklmnop_-=:") for _ in range(rng.randint(8, 16)))
    before = [rng.choice(_PARTS) for _ in range(rng.randint(0, 2))]
    after = [rng.choice(_PARTS) for _ in range(rng.randint(0, 2))]
    tail = rng.choice(["", "", "$", "\\b"])
    esc = "".join("\\" + c if c in "-=:" else c for c in lit)
    pattern = "".join(p for p, _ in before) + esc + "".join(p for p, _ in after) + tail

    def sample(r):
        s = "".join(f(r) for _, f in before) + lit + "".join(f(r) for _, f in after)
        roll = r.random()
        if roll < 0.15:  # break the literal
            i = r.randrange(len(s))
            s = s[:i] + r.choice("XY7") + s[i + 1:]
        elif roll < 0.3:  # cut it short
            s = s[: r.randrange(1, len(s))]
        return s

    return pattern, sample


def anchored_text(rng, samplers, nlines):
    filler = ["alpha", "beta", "status=200", "GET", "x", "foo", "bar7", "1234", "ZZ_TOP", "ab", "cd", ""]
    lines = []
    for _ in range(nlines):
        toks = [rng.choice(filler) for _ in range(rng.randint(0, 6))]
        for _ in range(rng.choice([0, 1, 1, 2])):
            toks.insert(rng.randint(0, len(toks)), rng.choice(samplers)(rng))
        sep = rng.choice([" ", " ", "", ","])
        lines.append(sep.join(toks))
    return ("\n".join(lines) + "\n").encode()


def py_flags(flags: int) -> int:
    import re

    f = 0
    if flags & 1:
        f |= re.I
    if flags & 2:
        f |= re.S
    if flags & 4:
        f |= re.M
    return f


# ---- independent pin of match END offsets (`to`): they decide SINGLEMATCH selection and the delivery order inside a line
# (hyperscanner.c:83-102 delivers in hs_scan's report order), and no reference fixture covers them
# (test_hypergrep.py:668-687 pins only one report per line).  Python `re` is the independent engine: end offset e is a
# match end iff some start s < e has a match of exactly [s, e) WITH the line's real context on both sides — a lookahead
# that pins the distance to the end of the line keeps `$`, `\b` and `\Z` honest at e (re.fullmatch(pos, endpos) would
# pretend the line ends at e), and `pos` keeps `^` / `\b` honest at s.
import re  # noqa: E402

_END_RE_CACHE: dict = {}


def ends_by_brute_force(pat: str, flags: int, line: bytes) -> list[int]:
    n = len(line)
    ends = []
    for e in range(1, n + 1):
        key = (pat, flags, n - e)
        cre = _END_RE_CACHE.get(key)
        if cre is None:
            cre = _END_RE_CACHE[key] = re.compile(b"(?:" + pat.encode() + b")(?=(?s:.{%d})\\Z)" % (n - e), py_flags(flags))
        if any(cre.match(line, s) for s in range(e)):
            ends.append(e)
    return ends


def split_pieces(data: bytes) -> list[bytes]:
    pieces = data.split(b"\n")
    return [p + b"\n" for p in pieces[:-1]] + ([pieces[-1]] if pieces[-1] else [])


def end_offset_cases(seed: int, per_seed: int = 16, accepts=None):
    """(pattern, flags, text, [(line, to)]) cases; accepts(pattern, flags) filters out what the engine under test rejects."""
    rng = random.Random(7000 + seed)
    made = 0
    for _ in range(400):
        if made == per_seed:
            break
        pat = random_pattern(rng)
        flags = rng.choice([6, 6, 7, 2, 4])  # all-matches mode (no SINGLEMATCH); caseless / non-multiline / no DOTALL variants
        try:
            re.compile(pat.encode(), py_flags(flags))
        except re.error:
            continue
        if accepts is not None and not accepts(pat, flags):
            continue
        data = random_text(rng, 12, maxlen=14, final_newline=rng.random() < 0.8)
        _END_RE_CACHE.clear()
        want = []  # (line, to)
        for i, line in enumerate(split_pieces(data)):
            want += [(i, e) for e in ends_by_brute_force(pat, flags, line)]
        made += 1
        yield pat, flags, data, want


