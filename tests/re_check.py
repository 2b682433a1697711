"""An independent checker for the benchmark workloads: Python's `re` on bytes, every expression on its own over the whole
buffer (existence of a match per (line, expression) is all HS_FLAG_SINGLEMATCH exposes).  Neither the
oracle nor the product is involved.  Only valid for expressions that cannot match across a newline (asserted)."""
from __future__ import annotations

import re

import numpy as np


def line_id_pairs(text: bytes, patterns, ids=None) -> set:
    """{(0-based line, id)}: lines in which the expression has a match, for every expression (flags DOTALL | MULTILINE)."""
    nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
    out = set()
    for i, p in enumerate(patterns):
        assert not re.search(r"\\s|\\n|\\W|\\D|\[\^|(?<!\\)\.", p), f"{p!r} could match across a newline: not a case for this checker"
        rx = re.compile(p.encode(), re.S | re.M)
        ident = ids[i] if ids else i
        for m in rx.finditer(text):
            assert b"\n" not in m.group()
            out.add((int(np.searchsorted(nl, m.start(), side="left")), ident))
    return out


def literal_line_id_pairs(text: bytes, literals, ids=None) -> set:
    """The same for a large set of plain literals (config 5), in one pass: an alternation of the escaped literals.  Valid when
    no occurrence of one literal can overlap an occurrence of another (finditer reports non-overlapping matches): asserted
    for literals of the form a distinct first byte followed by bytes that never equal it."""
    lits = [x.encode() if isinstance(x, str) else x for x in literals]
    first = {x[0] for x in lits}
    assert all(not (first & set(x[1:])) for x in lits), "occurrences could overlap"
    index = {x: (ids[i] if ids else i) for i, x in enumerate(lits)}
    nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
    rx = re.compile(b"|".join(re.escape(x) for x in sorted(lits, key=len, reverse=True)))
    return {(int(np.searchsorted(nl, m.start(), side="left")), index[m.group()]) for m in rx.finditer(text)}
