"""CPU tests of the product's pattern compiler and of the scalar device logic (hg_core.h / hg_post.h),
replayed on the host by tests/native/hostsim.cpp and compared with the oracle."""
from __future__ import annotations

import random
import re

import pytest

import hgsim_py
import oracle_py
import regex_gen
from test_oracle import ACCEPTED, REJECTED


def oracle_hits(data, patterns, flags=None, ids=None, buffer_size=262140):
    rc, hits, nlines = oracle_py.scan_buffer(data, patterns, flags=flags, ids=ids, buffer_size=buffer_size)
    assert rc == 0
    return sorted(hits), nlines


def sim_hits(data, patterns, flags=None, ids=None, buffer_size=262140):
    db = hgsim_py.Db(patterns, flags, ids)
    assert db.ok(), db.error
    hits, stats = db.scan(data, buffer_size)
    return sorted(hits), stats, db


@pytest.mark.parametrize("pat", REJECTED)
def test_rejects_like_oracle(pat):
    assert not hgsim_py.Db([pat]).ok()


@pytest.mark.parametrize("pat", ACCEPTED)
def test_accepts_like_oracle(pat):
    db = hgsim_py.Db([pat])
    assert db.ok(), db.error


def test_flag_and_anchor_rules():
    assert not hgsim_py.Db(["abc"], flags=[16]).ok()
    assert not hgsim_py.Db(["a^b"], flags=[10]).ok()
    assert not hgsim_py.Db(["a$b"], flags=[10]).ok()
    assert hgsim_py.Db(["^ab$"], flags=[10]).ok()
    assert hgsim_py.Db(["a^b"]).ok()


def test_tiers_and_factors():
    long_ones = ["needle_in_haystack", "user=[a-z0-9_]{4,12} status=5[0-9]{2}", "[a-z]+@[a-z]+",
                 "(alpha_long_one|beta_long_two)x", "(?i)CaseLessLiteral"]
    # required literals of 7 bytes and more: dword-aligned windows, one per residue mod 4
    db = hgsim_py.Db(long_ones)
    assert db.ok(), db.error
    assert [db.tier(i) for i in range(5)] == [0, 0, 1, 0, 0]
    info = db.info()
    # one case-insensitive literal among five: its windows are stored in every case variant and nothing is folded (round 3)
    assert info["nslow"] == 1 and info["fold_mask"] == 0
    assert info["nfactors"] >= 5 and 4 * info["nfactors"] < info["nwindows"] <= 4 * (info["nfactors"] - 1) + 4 * 16
    assert db.selfcheck()["byte_windows"] == 0
    # a 6-byte literal joins: the whole set switches to byte-aligned probing; every literal has >= 5 bytes, so a window
    # starts at every SECOND byte and each literal gets one window per residue mod 2
    db = hgsim_py.Db(["foobar"] + long_ones)
    assert db.ok(), db.error
    assert [db.tier(i) for i in range(6)] == [0, 0, 0, 1, 0, 0]
    info = db.info()
    assert info["nslow"] == 1 and info["nwindows"] == 2 * info["nfactors"]
    check = db.selfcheck()
    assert check["byte_windows"] == 2 and check["violations"] == 0
    # a 4-byte literal: a window at every byte, one window per literal
    db = hgsim_py.Db(["fail"] + long_ones)
    assert db.ok() and db.selfcheck()["byte_windows"] == 1 and db.info()["nwindows"] == db.info()["nfactors"]
    # 3-byte literals: the byte after the literal is enumerated (256 windows, 128 distinct under case folding)
    db = hgsim_py.Db(["foo", "barbaz"])
    assert db.ok(), db.error
    assert [db.tier(i) for i in range(2)] == [0, 0] and db.info()["nwindows"] == 257
    assert db.selfcheck()["violations"] == 0
    # 2-byte literals and expressions without a required literal stay on every line
    db = hgsim_py.Db(["ab", "[0-9]+x"])
    assert [db.tier(i) for i in range(2)] == [1, 1]


def test_short_literals_byte_aligned_windows():
    """Every alignment of short literals, at the edges of the text, of tiles and of 1 KiB rows, with look-alikes around."""
    pats = ["ERROR", "WARN", "foo", "(?i)Fail", "a\\.b", "panic: [a-z]+", "x=\\d+;"]
    flags = [14, 14, 10, 14, 14, 6, 14]
    ids = [0, 1, 2, 3, 4, 5, 6]
    db = hgsim_py.Db(pats, flags, ids)
    assert db.ok(), db.error
    assert db.selfcheck()["byte_windows"] == 1 and db.selfcheck()["violations"] == 0
    assert [db.tier(i) for i in range(7)] == [0] * 6 + [1]  # "x=" is too short: that one runs on every line
    rng = random.Random(77)
    words = [b"ERROR", b"WARN", b"foo", b"FAIL", b"fAiL", b"a.b", b"panic: oops", b"x=12;", b"ERRO", b"WAR", b"fo", b"fai", b"x=;", b"foofoo"]
    for trial in range(6):
        out = bytearray()
        while len(out) < 40000:
            line = bytearray()
            for _ in range(rng.randint(0, 6)):
                line += rng.choice(words) if rng.random() < 0.5 else bytes(rng.choice(b"abcdefoOrRE =.;0123") for _ in range(rng.randint(1, 9)))
                if rng.random() < 0.5:
                    line += b" "
            out += line + b"\n"
        # occurrences that straddle 1 KiB rows and the 16 KiB tile edge, and one that ends the text
        for at in (1021, 1022, 1023, 2045, 16381, 16382, 16383, 32765):
            out[at:at + 5] = b"ERROR"
            out[at + 3000:at + 3003] = b"foo"
        data = bytes(out[:39000 + trial]) + rng.choice([b"foo", b"WARN", b"ERROR\n", b"fo"])
        want, _ = oracle_hits(data, pats, flags, ids)
        got, stats = db.scan(data)
        assert sorted(got) == want, trial
        tuned = hgsim_py.Db(pats, flags, ids)
        assert tuned.tune(data[:20000]) == 0 and tuned.selfcheck()["violations"] == 0
        assert sorted(tuned.scan(data)[0]) == want, trial


def test_expressions_that_can_never_match():
    """Contradictory assertions leave an automaton without nodes (found by tools/fuzz_gpu.py: the GPU staged the NEXT
    pattern's tables for it).  Such an expression compiles, never matches, and does not disturb its neighbours."""
    pats = ["\\b\\Bc", "[^a]-$", "\\B\\b$(1){1}", "needle_long", "x\\b\\By"]
    flags = [7, 10, 15, 14, 6]
    ids = [2, 3, 2, 0, 1]
    db = hgsim_py.Db(pats, flags, ids)
    assert db.ok(), db.error
    assert db.info()["max_nw"] == 1
    data = b"abc c-\nccc x y 1\nneedle_long c\nx-\n" * 50
    want, _ = oracle_hits(data, pats, flags, ids)
    got, _ = db.scan(data)
    assert sorted(got) == want and {h[1] for h in want} == {0, 3}


def test_five_byte_literals_probe_every_second_byte():
    """Sets whose literals all have >= 5 bytes: byte-aligned probing at even offsets only, two windows per literal; occurrences
    at every alignment, across rows / tiles and at the end of the text."""
    pats = ["ERROR", "panic", "(?i)Failed", "denied: [a-z]+", "needle_in_haystack", "x=\\d+;"]
    flags = [14, 14, 14, 6, 14, 14]
    ids = [0, 1, 2, 3, 4, 5]
    db = hgsim_py.Db(pats, flags, ids)
    assert db.ok(), db.error
    assert db.selfcheck()["byte_windows"] == 2 and db.selfcheck()["violations"] == 0
    assert [db.tier(i) for i in range(6)] == [0, 0, 0, 0, 0, 1]
    rng = random.Random(91)
    words = [b"ERROR", b"panic", b"FAILED", b"failed", b"denied: abc", b"needle_in_haystack", b"x=12;", b"ERRO", b"pani", b"faile", b"denied:", b"ERRORERROR"]
    for trial in range(6):
        out = bytearray()
        while len(out) < 40000:
            line = bytearray()
            for _ in range(rng.randint(0, 6)):
                line += rng.choice(words) if rng.random() < 0.5 else bytes(rng.choice(b"abcdefoOrRE =.;0123") for _ in range(rng.randint(1, 9)))
                if rng.random() < 0.5:
                    line += b" "
            out += line + b"\n"
        for at in (1019, 1020, 1021, 1022, 1023, 2045, 16379, 16380, 16381, 16382, 16383, 32765):
            out[at:at + 5] = b"ERROR"
            out[at + 3000:at + 3005] = b"panic"
        data = bytes(out[:39000 + trial]) + rng.choice([b"panic", b"ERROR", b"ERROR\n", b"pani"])
        want, _ = oracle_hits(data, pats, flags, ids)
        got, _ = db.scan(data)
        assert sorted(got) == want, trial
        tuned = hgsim_py.Db(pats, flags, ids)
        assert tuned.tune(data[:20000]) == 0 and tuned.selfcheck()["violations"] == 0
        assert sorted(tuned.scan(data)[0]) == want, trial


def test_many_three_byte_literals_use_three_byte_windows():
    """More 3-byte literals than the filter can hold with the byte after each enumerated: every window shrinks to 3 bytes
    (byte-aligned probing, hashes that ignore the dword's top byte) and no pattern falls back to the always-on tier."""
    rng = random.Random(31)
    alphabet = "abcdefghijklmnopqrstuvwxyz"
    words = sorted({"".join(rng.choice(alphabet) for _ in range(3)) for _ in range(400)})
    pats = words + ["needle", "(?i)MiXeD", "ab[0-9]x", "tail$"]
    flags = [14] * len(words) + [14, 14, 6, 14]
    ids = list(range(len(pats)))
    db = hgsim_py.Db(pats, flags, ids)
    assert db.ok(), db.error
    info, check = db.info(), db.selfcheck()
    assert check["byte_windows"] == 1 and check["violations"] == 0
    assert info["nslow"] == 0 and info["nwindows"] == info["nfactors"]  # one 3-byte window per literal, nothing enumerated
    for trial in range(3):
        out = bytearray()
        while len(out) < 50000:
            toks = [rng.choice(words + ["needle", "mixed", "MIXED", "ab7x", "tail", "zz", "q"]) if rng.random() < 0.3
                    else "".join(rng.choice(alphabet + "  .=") for _ in range(rng.randint(1, 7))) for _ in range(rng.randint(0, 12))]
            out += (" ".join(toks) + "\n").encode()
        for at in (1021, 1022, 1023, 16381, 16382, 16383, 32766):
            out[at:at + 3] = words[at % len(words)].encode()
        data = bytes(out[:49000 + trial]) + words[trial].encode()[: 2 + (trial & 1)]
        want, _ = oracle_hits(data, pats, flags, ids)
        got, _ = db.scan(data)
        assert sorted(got) == want, trial
    tuned = hgsim_py.Db(pats, flags, ids)
    assert tuned.tune(data[:30000]) == 0 and tuned.selfcheck()["violations"] == 0
    assert sorted(tuned.scan(data)[0]) == want


@pytest.mark.parametrize("seed", range(40))
def test_random_patterns_match_oracle(seed):
    rng = random.Random(5000 + seed)
    done = 0
    for _ in range(40):
        k = rng.randint(1, 3)
        pats = [regex_gen.random_pattern(rng) for _ in range(k)]
        flags = [rng.choice([14, 14, 15, 10, 6, 12]) for _ in range(k)]
        ids = [rng.randint(0, 2) for _ in range(k)]
        if oracle_py.check_patterns(pats, flags=flags) != 0:
            continue
        db = hgsim_py.Db(pats, flags, ids)
        if not db.ok():
            continue  # documented frontier differences are checked elsewhere
        data = regex_gen.random_text(rng, 40, final_newline=rng.random() < 0.8)
        want, nlines = oracle_hits(data, pats, flags, ids)
        got, stats = db.scan(data)
        assert sorted(got) == want, (pats, flags, ids, data)
        done += 1
    assert done > 10


def _log_text(rng, nlines, needles, p_hit=0.2, maxlen=120):
    words = ["alpha", "beta", "gamma", "delta", "status=200", "user=bob", "GET", "/index.html", "10.0.0.1", "ok",
             "warn", "retry", "timeout=30", "id=12345", "x"]
    out = []
    for _ in range(nlines):
        n = rng.randint(0, 12)
        toks = [rng.choice(words) for _ in range(n)]
        if needles and rng.random() < p_hit:
            toks.insert(rng.randint(0, len(toks)), rng.choice(needles))
        out.append(" ".join(toks)[:maxlen])
    return ("\n".join(out) + "\n").encode()


def test_literal_anchored_tier_matches_oracle():
    rng = random.Random(77)
    pats = ["needle_in_haystack", "ERR_DISK_FULL_[0-9]{3}", "user=[a-z0-9_]{4,12} status=5[0-9]{2}",
            "(?i)caseless_needle", "(first_long_alt|second_long_alt) tail", "connection reset by peer$",
            "^kernel panic -", "\\bwordbound_token\\b"]
    needles = ["needle_in_haystack", "ERR_DISK_FULL_042", "ERR_DISK_FULL_04", "user=alice_01 status=503",
               "user=al status=503", "CaseLess_Needle", "first_long_alt tail", "second_long_alt tail",
               "second_long_alt  tail", "connection reset by peer", "kernel panic - not syncing",
               "wordbound_token", "xwordbound_tokenx", "needle_in_haystac"]
    data = _log_text(rng, 3000, needles)
    for ids in (None, list(range(len(pats)))):
        want, nlines = oracle_hits(data, pats, ids=ids)
        got, stats, db = sim_hits(data, pats, ids=ids)
        assert all(db.tier(i) == 0 for i in range(len(pats)))
        assert got == want
        assert stats["pieces"] == nlines
        assert len(want) > 100


def test_tile_boundaries_and_long_lines():
    rng = random.Random(3)
    pat = ["needle_in_haystack", "tail_anchor_zz$"]
    # lines whose literals straddle 16 KiB tile boundaries, lines longer than a tile, and a line longer than bs1
    chunks = []
    pos = 0
    for i in range(40):
        pad = rng.randint(0, 9000)
        line = b"a" * pad + b" needle_in_haystack " + b"b" * rng.randint(0, 9000) + b" tail_anchor_zz\n"
        chunks.append(line)
        pos += len(line)
    chunks.append(b"q" * 70000 + b"needle_in_haystack" + b"r" * 70000 + b"tail_anchor_zz\n")
    chunks.append(b"short needle_in_haystack\n")
    chunks.append(b"no newline at end needle_in_haystack tail_anchor_zz")
    data = b"".join(chunks)
    for bs in (262140, 20001, 16385):
        want, nlines = oracle_hits(data, pat, ids=[0, 1], buffer_size=bs)
        got, stats, _ = sim_hits(data, pat, ids=[0, 1], buffer_size=bs)
        assert got == want, bs
        assert stats["pieces"] == nlines


def test_straddle_every_alignment():
    lit = "needle_in_haystack"
    for shift in range(0, 40):
        data = b"x" * (16384 - 20 + shift) + lit.encode() + b"\nnext line\n"
        want, _ = oracle_hits(data, [lit])
        got, _, _ = sim_hits(data, [lit])
        assert got == want and len(got) == 1


def test_nul_rules_match_oracle():
    pats = ["needle_in_haystack", "x"]
    data = (b"\0\0needle_in_haystack\n" b"ab\0needle_in_haystack x\n" b"x\0\0\n" b"\0\n" b"needle_in_haystack\0x\n"
            b"\0\0\0")
    for ids in (None, [1, 2]):
        want, nlines = oracle_hits(data, pats, ids=ids)
        got, stats, _ = sim_hits(data, pats, ids=ids)
        assert got == want
        assert stats["pieces"] == nlines


def test_not_singlematch_and_mixed_ids():
    data = b"aaa needle_in_haystack needle_in_haystack\nba\n"
    pats = ["a", "needle_in_haystack", "needle"]
    flags = [6, 6, 14]
    ids = [0, 1, 1]
    want, _ = oracle_hits(data, pats, flags, ids)
    got, _, _ = sim_hits(data, pats, flags, ids)
    assert got == want


def test_many_patterns_share_windows():
    rng = random.Random(11)
    lits = ["tok_%04x_%s" % (i, "".join(rng.choice("abcdef") for _ in range(rng.randint(2, 10)))) for i in range(300)]
    data = _log_text(rng, 2000, lits, p_hit=0.3)
    want, _ = oracle_hits(data, lits, ids=list(range(len(lits))))
    got, stats, db = sim_hits(data, lits, ids=list(range(len(lits))))
    assert got == want and len(got) > 300
    assert db.info()["nslow"] == 0


def test_large_literal_set_uses_the_wide_filter():
    """BASELINE config 5: 4096 literals = 16384 windows, more than the one-fingerprint-per-slot filter holds."""
    from hypergrep_amd import benchspec, device

    pats, needles, hpm = benchspec.c5_spec()
    ids = list(range(len(pats)))
    data = device.synth_host(96 << 10, benchspec.SEED_BASE + 5, needles, hpm)
    want, _ = oracle_hits(data, pats, ids=ids)
    got, stats, db = sim_hits(data, pats, ids=ids)
    assert db.info()["nwindows"] == 4 * len(pats) and db.info()["nslow"] == 0
    assert got == want and len(got) > 50


def test_small_buffer_sizes_match_oracle():
    rng = random.Random(21)
    pats = ["needle_in_haystack", "x", "ab+c"]
    chunks = []
    for _ in range(300):
        n = rng.randint(0, 60)
        line = "".join(rng.choice("abcx _-") for _ in range(n))
        if rng.random() < 0.2:
            line += " needle_in_haystack "
        chunks.append(line.encode() + b"\n")
    chunks.append(b"y" * 40000 + b"needle_in_haystack" + b"x" * 300 + b"\n")
    chunks.append(b"tail without newline x")
    data = b"".join(chunks)
    for bs in (2, 3, 8, 9, 33, 100, 4097, 16384):
        for ids in (None, [5, 6, 7]):
            want, nlines = oracle_hits(data, pats, ids=ids, buffer_size=bs)
            got, stats, _ = sim_hits(data, pats, ids=ids, buffer_size=bs)
            assert got == want, (bs, ids)
            assert stats["pieces"] == nlines, bs


def test_golden_plumbing_vectors_through_device_logic():
    import base64, json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "plumbing_vectors.json")) as f:
        vectors = json.load(f)
    for v in vectors:
        if v["data"] is None or "max_match_count" in v["kwargs"]:
            continue
        data = base64.b64decode(v["data"])
        kw = v["kwargs"]
        got, _, _ = sim_hits(data, v["patterns"], flags=kw.get("flags"), ids=kw.get("ids"), buffer_size=kw.get("buffer_size", 262140))
        want = sorted((r[0], r[1], base64.b64decode(r[2])) for r in v["rows"])
        assert sorted((h[0], h[1], data[h[3]:h[3] + h[4]]) for h in got) == want, v["name"]


@pytest.mark.parametrize("n_literals, seed", [(40, 1), (600, 2), (3000, 3), (9000, 4)])
def test_prefilter_tables_never_lose_a_window(n_literals, seed):
    """Every window of every literal passes its filter slot, its slot's second level with the literal's own neighbours,
    and the verify pass's discriminated bucket — before and after tuning on text that contains look-alikes."""
    rng = random.Random(seed)
    alphabet = "abcdefghijklmnopqrstuvwxyz0123456789_-=/"
    lits = sorted({"".join(rng.choice(alphabet) for _ in range(rng.randint(7, 20))) for _ in range(n_literals)})
    pats = [re.escape(l) for l in lits] + ["user=[a-z0-9_]{4,12} status=5[0-9]{2}", "(?i)CaseLess_Literal_[0-9]+", "prefix_(?:alpha|beta|gamma)_suffix"]
    db = hgsim_py.Db(pats, ids=list(range(len(pats))))
    assert db.ok(), db.error
    first = db.selfcheck()
    assert first["violations"] == 0
    sample = ("\n".join(rng.choice(lits)[:-1] + "x status=200 user=abcd" for _ in range(4000)) + "\n").encode()
    assert db.tune(sample) == 0
    second = db.selfcheck()
    assert second["violations"] == 0
    if n_literals >= 3000:
        assert second["wide"] or second["crowded_slots"] >= 0  # large sets: either mode is fine, the invariants are what counts


@pytest.mark.parametrize("seed", range(8))
def test_random_literal_anchored_patterns_match_oracle(seed):
    """Random expressions around a long literal (prefilter tier, every confirm mode) on text made of their own matches
    and near-misses: compiler + scalar device logic vs the oracle."""
    rng = random.Random(4100 + seed)
    pairs = [regex_gen.anchored_pattern(rng) for _ in range(rng.randint(3, 10))]
    pats = [p for p, _ in pairs]
    flags = [rng.choice([14, 14, 15, 6, 10]) for _ in pats]
    ids = [rng.randint(0, 3) for _ in pats]
    assert oracle_py.check_patterns(pats, flags=flags) == 0, pats
    data = regex_gen.anchored_text(rng, [s for _, s in pairs], 600)
    want, _ = oracle_hits(data, pats, flags, ids)
    got, _, db = sim_hits(data, pats, flags, ids)
    assert got == want, (pats, flags, ids)
    assert len(want) > 50 and db.info()["nslow"] == 0


def test_windowed_confirm_lead_of_the_required_literal():
    """SINGLEMATCH automata are confirmed on a window: matches that start at most `lit_lead` bytes before the verified
    occurrence of the required literal (hg_core.h hg_confirm_window, the host mirror of hg_confirm_dev.h).  Expressions
    whose literal follows optional, alternative, bounded and unbounded prefixes, repeated groups holding the literal,
    assertions either side, several occurrences per line, overlapping matches: the smallest end per line must not change."""
    rng = random.Random(515)
    pats = ["[a-z]{0,3}needle_alpha_1", "(?:ab|cde)?needle_beta_22 tail", "x*needle_gamma_3+", "(?:aa|b)+needle_delta_4", "\\b[0-9]{2,5}-needle_eps_5\\b",
            "(?:pre_needle_zeta_6|needle_zeta_6x)[0-9]", "q?(?:needle_eta_77){1,3}z", "[A-Z][a-z]+ needle_theta_8$", "^.{0,6}needle_iota_9", "(?i)u{2,}Needle_Kappa_10",
            "(?:foo(?:needle_lambda_11|bar_needle_mu_12)){1,2}!", "a.c.e needle_nu_13"]
    flags = [14] * len(pats)
    flags[8] = 10
    assert oracle_py.check_patterns(pats, flags=flags) == 0
    pieces = ["needle_alpha_1", "abcneedle_alpha_1", "zzzzzneedle_alpha_1 needle_alpha_1", "abneedle_beta_22 tail", "cdeneedle_beta_22 tail", "needle_beta_22 tai",
              "xxxxxxneedle_gamma_3333", "needle_gamma_", "aabaaneedle_delta_4", "abneedle_delta_4", "12-needle_eps_5", "123456-needle_eps_5", "x12-needle_eps_5", "99-needle_eps_5x",
              "pre_needle_zeta_67", "needle_zeta_6x8", "needle_zeta_68", "qneedle_eta_77needle_eta_77z", "needle_eta_77needle_eta_77needle_eta_77needle_eta_77z",
              "Hello needle_theta_8", "Hello needle_theta_8 more", "abcneedle_iota_9", "abcdefgneedle_iota_9", "uuuuNEEDLE_kappa_10", "uneedle_kappa_10",
              "fooneedle_lambda_11foobar_needle_mu_12!", "foobar_needle_mu_12!", "fooneedle_lambda_11", "abcde needle_nu_13", "a c e needle_nu_13", "ab de needle_nu_13"]
    lines = []
    for _ in range(2500):
        n = rng.randint(1, 4)
        lines.append(" ".join(rng.choice(pieces) if rng.random() < 0.6 else "".join(rng.choice("abcxuq 019-AZ") for _ in range(rng.randint(0, 12))) for _ in range(n)))
    data = ("\n".join(lines) + "\n").encode()
    for ids in (list(range(len(pats))), [0] * len(pats)):
        want, nlines = oracle_hits(data, pats, flags, ids)
        got, stats, db = sim_hits(data, pats, flags, ids)
        assert all(db.tier(i) == 0 for i in range(len(pats)))
        assert got == want
        assert len(want) > 1500


# ---- match END offsets of the product's automata against the Python-`re` brute force (regex_gen.ends_by_brute_force):
# independent of the oracle, which the same brute force pins in tests/test_oracle.py
@pytest.mark.parametrize("seed", range(20))
def test_end_offsets_against_python_re(seed):
    nonempty = 0
    for pat, flags, data, want in regex_gen.end_offset_cases(seed, accepts=lambda p, f: hgsim_py.Db([p], [f]).ok()):
        got, _, _ = sim_hits(data, [pat], [flags])
        assert [(h[0], h[2]) for h in got] == want, (pat, flags, data)
        got1, _, _ = sim_hits(data, [pat], [flags | 8])  # SINGLEMATCH: the smallest end offset of each line
        first = {}
        for line, to in want:
            first.setdefault(line, to)
        assert [(h[0], h[2]) for h in got1] == sorted(first.items()), (pat, flags, data)
        nonempty += bool(want)
    assert nonempty >= 2


def test_direct_window_table_on_the_benchmark_sets():
    """Round 3's direct window table (hg_db.h HgWinBucket): every window of every literal is found in it (selfcheck invariant 4),
    config 5's K%04x-%08x literals get single-owner windows (the selection avoids the `xxx-` windows sixteen literals share) and
    the verify pass asks the table first there; config 3's class expressions share their stems: the discriminated buckets first."""
    from hypergrep_amd import benchspec, device

    for name, first, most_shared in (("c5", 1, 400), ("c3", 0, 400), ("c1", 1, 0)):
        pats, needles, hpm = getattr(benchspec, name + "_spec")()
        db = hgsim_py.Db(pats, None, list(range(len(pats))))
        assert db.ok(), db.error
        check = db.selfcheck()
        assert check["violations"] == 0 and check["table_first"] == first and check["shared_windows"] <= most_shared, (name, check)
        sample = device.synth_host(1 << 20, benchspec.SEED_BASE + int(name[1]), needles, hpm)
        assert db.tune(sample) == 0
        tuned = db.selfcheck()
        assert tuned["violations"] == 0 and tuned["table_first"] == first, (name, tuned)


def test_case_insensitive_literals_as_case_variants_or_folded(monkeypatch):
    """A set with a few case-insensitive literals stores their windows in every case variant and folds nothing (round 3); with many
    (or with HG_NO_CASE_EXPAND) the text is folded as before.  Same reports either way, equal to the oracle's, on text with every
    mix of cases, look-alikes that only differ by the fold bit ('@' / '`', '[' / '{') and discriminated window groups."""
    rng = random.Random(11)
    pats = ["(?i)Unhandled_Exception_9: [a-z]+error", "needle_in_haystack", "(?i)tls_handshake_failed", "status=5[0-9]{2} retry",
            "(?i)status=4[0-9]{2} GIVE_UP", "(?i)abc@def\\[xyz", "plain_literal_{x}"]
    flags = [14, 14, 14, 14, 15, 14, 14]
    ids = list(range(len(pats)))
    samples = [b"Unhandled_Exception_9: fooerror", b"UNHANDLED_exception_9: barERROR", b"unhandled_exception_9: xerror", b"Unhandled_Exception_8: fooerror",
               b"TLS_HANDSHAKE_FAILED", b"tls_Handshake_Failed", b"tls_handshake_failex", b"needle_in_haystack", b"NEEDLE_IN_HAYSTACK", b"status=503 retry",
               b"STATUS=503 retry", b"status=404 give_up", b"Status=404 GIVE_UP", b"ABC@DEF[XYZ", b"abc`def{xyz", b"abc@def[xyz", b"plain_literal_{x}", b"plain_literal_[x]"]
    lines = []
    for _ in range(1500):
        line = bytearray()
        for _ in range(rng.randint(0, 4)):
            line += rng.choice(samples) if rng.random() < 0.5 else bytes(rng.choice(b"abcXYZ_ =@`[{19") for _ in range(rng.randint(1, 12)))
            line += b" "
        lines.append(bytes(line))
    data = b"\n".join(lines) + b"\n"
    want, nlines = oracle_hits(data, pats, flags, ids)
    assert len(want) > 300
    for knob, fold in ((None, 0), ("1", 0x20202020)):
        if knob:
            monkeypatch.setenv("HG_NO_CASE_EXPAND", knob)
        db = hgsim_py.Db(pats, flags, ids)
        assert db.ok(), db.error
        assert db.info()["fold_mask"] == fold and db.selfcheck()["violations"] == 0
        got, stats = db.scan(data)
        assert sorted(got) == want and stats["pieces"] == nlines, knob
        assert db.tune(data[:40000]) == 0 and db.selfcheck()["violations"] == 0
        got, _ = db.scan(data)
        assert sorted(got) == want, (knob, "tuned")
    # more than 64 case-insensitive literals: folding
    many = [f"(?i)Keyword_{i:03d}_Tail" for i in range(70)]
    db = hgsim_py.Db(many)
    assert db.ok() and db.info()["fold_mask"] == 0x20202020
