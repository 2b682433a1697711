"""CPU tests: the oracle against the committed golden fixtures and against Python `re`.

Golden fixtures come from tests/golden/make_golden.py (reference shim + reference Python over the
oracle's libhs face, expectations from the reference's own test tables).
"""
from __future__ import annotations

import base64
import json
import os
import random
import re

import pytest

import oracle_py
import regex_gen

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
FILES = os.path.join(GOLD, "files")

with open(os.path.join(GOLD, "plumbing_vectors.json"), encoding="utf-8") as _f:
    VECTORS = json.load(_f)
with open(os.path.join(GOLD, "reference_tables.json"), encoding="utf-8") as _f:
    TABLES = json.load(_f)


def _kw(v):
    kw = dict(v["kwargs"])
    return kw


@pytest.mark.parametrize("v", [v for v in VECTORS if v["data"] is not None], ids=lambda v: v["name"])
def test_plumbing_vectors_file_and_buffer(v, tmp_path):
    data = base64.b64decode(v["data"])
    kw = _kw(v)
    path = tmp_path / "in.txt"
    path.write_bytes(data)
    rc, rows, batches = oracle_py.scan_file(str(path), v["patterns"], **kw)
    want = [(r[0], r[1], base64.b64decode(r[2])) for r in v["rows"]]
    assert rc == v["rc"]
    assert rows == want
    assert batches == v["batches"]
    # buffer API (restated gzgets splitter) must agree with the zlib-gzgets file path
    kw.pop("buffer_count", None)
    rc2, hits, _ = oracle_py.scan_buffer(data, v["patterns"], **kw)
    assert rc2 == v["rc"]
    assert [(h[0], h[1], data[h[3]:h[3] + h[4]]) for h in hits] == want


def test_missing_file_rc6():
    v = next(v for v in VECTORS if v["name"] == "K_missing_file")
    rc, rows, batches = oracle_py.scan_file("/nonexistent/definitely/missing", ["x"])
    assert (rc, rows, batches) == (v["rc"], [], [])


@pytest.mark.parametrize("case", TABLES["check_compatibility"], ids=lambda c: c["name"])
def test_reference_table_check(case):
    assert oracle_py.check_patterns(case["patterns"]) == case["returns"]


@pytest.mark.parametrize("case", TABLES["scan"], ids=lambda c: c["name"])
def test_reference_table_scan(case):
    rc, rows, _ = oracle_py.scan_file(os.path.join(FILES, case["file"]), case["patterns"])
    assert rc == 0
    assert [f"{ln}:{line.decode(errors='ignore').rstrip()}" for ln, _id, line in rows] == case["returns"]


@pytest.mark.parametrize("case", [c for c in TABLES["grep"] if "returns" in c and c["returns"][1] == 0],
                         ids=lambda c: c["name"])
def test_reference_table_grep(case):
    rc, rows, _ = oracle_py.scan_file(os.path.join(FILES, case["file"]), case["patterns"])
    assert rc == 0
    assert [[ln + 1, line.decode()] for ln, _id, line in rows] == case["returns"][0]


REJECTED = [
    "(?<!foo)bar", "(?<=foo)bar", "foo(?=bar)", "foo(?!bar)", "(a)\\1", "(?>a+)b", "a*+", "a++", "(?(1)a|b)",
    "(?R)", "\\Gabc", "a\\Kb", "\\X", "\\R", "\\p{L}", "a*", "a?", "(a|b*)", "^", "$", "\\b", "(?:)", "a{3,2}",
    "a{40000}", "a{32768}", "(", ")", "a)", "[a", "[z-a]", "*a", "a**b" if False else "+a", "\\", "x{2,1}",
]
ACCEPTED = [
    "foobar", "fo{2}bar", "fo+bar", "barfoo\\+", "a|b", "(a|b)c", "[a-z0-9_]{4,12}", "user=[a-z0-9_]{4,12} status=5[0-9]{2}",
    "^abc", "abc$", "\\bfoo\\b", "a.c", "(?i)abc", "(?i:a)b", "\\x41\\x{42}", "\\Qa.b\\E", "a{2}", "a{,3}", "{a", "a{x",
    "(?P<n>a)", "(?<n>a)", "(?#c)a", "[[:alpha:]]+", "[\\]a]", "[]a]", "[^]a]", "\\d+\\.\\d+", "a??b", "a*?b", "a+?",
    "\\101", "\\0", "[\\d-z]", "\\ca", "a\\z", "a\\Z", "\\Aa",
    # beyond 1024 automaton positions (Hyperscan's bounded-repeat limit is 32767; more of them in tests/huge_cases.py)
    "[a-z]{2000}x", ".{0,3000}foo", "foo.{0,3000}bar", "(abc|def){200}", "a{32767}",
]


@pytest.mark.parametrize("pat", REJECTED)
def test_rejected(pat):
    assert oracle_py.check_patterns([pat]) == 4


@pytest.mark.parametrize("pat", ACCEPTED)
def test_accepted(pat):
    assert oracle_py.check_patterns([pat]) == 0


def test_embedded_anchor_rules():
    nonml = [2 | 8]
    assert oracle_py.check_patterns(["a^b"], flags=nonml) == 4
    assert oracle_py.check_patterns(["a$b"], flags=nonml) == 4
    assert oracle_py.check_patterns(["^ab$"], flags=nonml) == 0
    assert oracle_py.check_patterns(["a^b"]) == 0  # multiline: allowed
    assert oracle_py.check_patterns(["abc"], flags=[16]) == 4  # unsupported flag bits


def _py_flags(flags: int) -> int:
    f = 0
    if flags & 1:
        f |= re.I
    if flags & 2:
        f |= re.S
    if flags & 4:
        f |= re.M
    return f


@pytest.mark.parametrize("seed", range(40))
def test_python_re_crosscheck(seed):
    rng = random.Random(1000 + seed)
    checked = 0
    for _ in range(60):
        pat = regex_gen.random_pattern(rng)
        flags = rng.choice([14, 14, 14, 15, 10, 12, 8])
        try:
            cre = re.compile(pat.encode(), _py_flags(flags))
        except re.error:
            continue
        if oracle_py.check_patterns([pat], flags=[flags]) != 0:
            continue  # empty-matchable / embedded anchors: rejected, as Hyperscan does
        data = regex_gen.random_text(rng, 30, final_newline=rng.random() < 0.8)
        rc, hits, _ = oracle_py.scan_buffer(data, [pat], flags=[flags])
        assert rc == 0
        got = {h[0] for h in hits}
        want = set()
        pieces = data.split(b"\n")
        lines = [p + b"\n" for p in pieces[:-1]] + ([pieces[-1]] if pieces[-1] else [])
        for i, line in enumerate(lines):
            m = cre.search(line)
            # Hyperscan never reports empty matches; a non-empty match must exist
            if m is not None and any(mm.end() > mm.start() for mm in cre.finditer(line)):
                want.add(i)
        assert got == want, (pat, flags, data)
        checked += 1
    assert checked > 20


def test_min_end_offset_and_all_ends():
    # SINGLEMATCH reports the smallest end offset; without it every distinct end offset is reported
    rc, hits, _ = oracle_py.scan_buffer(b"xxabcabc\n", ["abc", "bc"], ids=[1, 2])
    assert [(h[0], h[1], h[2]) for h in hits] == [(0, 1, 5), (0, 2, 5)]
    rc, hits, _ = oracle_py.scan_buffer(b"xxabcabc\n", ["abc"], flags=[6])
    assert [(h[0], h[1], h[2]) for h in hits] == [(0, 0, 5), (0, 0, 8)]
    rc, hits, _ = oracle_py.scan_buffer(b"foobar\n", ["foo.*"], flags=[6])
    assert [h[2] for h in hits] == [3, 4, 5, 6, 7]


def test_dollar_and_boundaries():
    def lines(data, pat, flags=14):
        return [h[0] for h in oracle_py.scan_buffer(data, [pat], flags=[flags])[1]]

    assert lines(b"foo\nfoo bar\nbarfoo", "foo$") == [0, 2]
    assert lines(b"foo\nfoo bar\n", "foo$", flags=10) == [0]  # non-multiline: end / before final \n of the piece
    assert lines(b"a foo\nfoo b\n", "^foo") == [1]
    assert lines(b"foo_x\nfoo x\nxfoo\n", "\\bfoo\\b") == [1]
    assert lines(b"ab\n", "b\\n") == [0]  # the trailing newline is part of the scanned line
    assert lines(b"ab", "b\\n") == []


# ---- independent pin of match END offsets (`to`): brute force over Python `re` (tests/regex_gen.py::ends_by_brute_force)
@pytest.mark.parametrize("seed", range(20))
def test_end_offsets_against_python_re(seed):
    nonempty = 0
    for pat, flags, data, want in regex_gen.end_offset_cases(seed, accepts=lambda p, f: oracle_py.check_patterns([p], flags=[f]) == 0):
        rc, hits, _ = oracle_py.scan_buffer(data, [pat], flags=[flags])
        assert rc == 0
        assert [(h[0], h[2]) for h in hits] == want, (pat, flags, data)  # every distinct end offset, ascending inside a line
        # SINGLEMATCH: one report per line, the smallest end offset
        rc, hits1, _ = oracle_py.scan_buffer(data, [pat], flags=[flags | 8])
        first = {}
        for line, to in want:
            first.setdefault(line, to)
        assert [(h[0], h[2]) for h in hits1] == sorted(first.items()), (pat, flags, data)
        nonempty += bool(want)
    assert nonempty >= 2


def test_end_offsets_known_answers():
    # hand-checked: the brute force itself (so that a bug in it cannot hide behind agreement with the oracle)
    assert regex_gen.ends_by_brute_force("abc", 6, b"xxabcabc\n") == [5, 8]
    assert regex_gen.ends_by_brute_force("foo.*", 6, b"foobar\n") == [3, 4, 5, 6, 7]
    assert regex_gen.ends_by_brute_force("foo$", 6, b"foo\n") == [3]          # `$` before the final newline only
    assert regex_gen.ends_by_brute_force("o\\b", 6, b"foo bar\n") == [3]       # `\b` looks at the byte after the match
    assert regex_gen.ends_by_brute_force("a+", 6, b"caab\n") == [2, 3]
    assert regex_gen.ends_by_brute_force("^b", 6, b"ab\n") == []
