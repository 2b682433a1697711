"""Expressions beyond 1024 automaton positions: the compiler accepts what the oracle (and Hyperscan: bounded repeats up to
32767) accepts, and the sparse tables + the host mirror of the wave-cooperative routine (hg_core.h hg_huge_scan_slice)
reproduce the oracle's reports.  The GPU side of the same cases: test_gpu_parity.py::test_huge_patterns_*."""
from __future__ import annotations

import pytest

import hgsim_py
import oracle_py
from huge_cases import ACCEPTED_HUGE, REJECTED_HUGE, SCAN_CASES, case_text


@pytest.mark.parametrize("pat", ACCEPTED_HUGE)
def test_accepted_by_oracle_and_product(pat):
    assert oracle_py.check_patterns([pat]) == 0
    db = hgsim_py.Db([pat])
    assert db.ok(), db.error
    assert hgsim_py.lib().hgsim_pattern_nodes(db.h, 0) > 1024


@pytest.mark.parametrize("pat", REJECTED_HUGE)
def test_rejected_by_oracle_and_product(pat):
    assert oracle_py.check_patterns([pat]) == 4
    assert not hgsim_py.Db([pat]).ok()


def test_program_size_frontier_matches_the_oracle():
    """Both bound an expression by its Thompson program size (400 000): the frontier is the same expression."""
    for n, want in ((399999, 0), (400000, 0), (400001, 4)):
        # a{32767} x 12 = 393204 instructions, the rest as single characters
        pat = "a{32767}" * 12 + "b" * (n - 12 * 32767)
        assert oracle_py.check_patterns([pat]) == want, n
        assert hgsim_py.Db([pat]).ok() == (want == 0), n


@pytest.mark.parametrize("case", range(len(SCAN_CASES)))
def test_scan_matches_oracle(case):
    pats, flags, ids, kind = SCAN_CASES[case]
    db = hgsim_py.Db(pats, flags, ids)
    assert db.ok(), db.error
    for seed in ((1,) if kind == "a32767" else (1, 2)):  # (the oracle's Pike VM needs seconds for each 32767-long run)
        data = case_text(kind, seed)
        rc, want, nlines = oracle_py.scan_buffer(data, pats, flags=flags, ids=ids)
        assert rc == 0
        got, stats = db.scan(data)
        assert sorted(got) == sorted(want), (pats, seed)
        assert stats["pieces"] == nlines
        assert want, "the case must have hits"


def test_small_scan_buffer_splits_lines_inside_a_huge_match():
    pats, data = ["foo.{0,3000}bar", "[a-z]{2000}x"], case_text("dotfoo", 3) + case_text("az2000x", 3)
    db = hgsim_py.Db(pats, None, [0, 1])
    for bs in (100, 2500, 4096):
        rc, want, nlines = oracle_py.scan_buffer(data, pats, ids=[0, 1], buffer_size=bs)
        got, stats = db.scan(data, bs)
        assert rc == 0 and sorted(got) == sorted(want) and stats["pieces"] == nlines, bs
