"""GPU parity tests (run on the MI355X box with `-m gpu`): every case goes through the C ABI of
hypergrep_amd/lib/libhyperscanner.so — Face B `hyperscan()` via the Python API mirror, or the hg_* device-buffer
API — and is compared bit-exactly with the oracle and the committed golden fixtures.

Nothing here reads /root/reference.
"""
from __future__ import annotations

import base64
import json
import os
import random

import pytest

import oracle_py
import regex_gen

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
FILES = os.path.join(GOLD, "files")

with open(os.path.join(GOLD, "plumbing_vectors.json"), encoding="utf-8") as _f:
    VECTORS = json.load(_f)
with open(os.path.join(GOLD, "reference_tables.json"), encoding="utf-8") as _f:
    TABLES = json.load(_f)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the product has no CPU path to fall back to")
    return torch


def _loaded_native() -> bool:
    with open("/proc/self/maps", encoding="utf-8") as maps:
        return "hypergrep_amd/lib/libhyperscanner.so" in maps.read()


def gpu_scan_buffer(torch, data: bytes, patterns, flags=None, ids=None, buffer_size=262140, line_base=0):
    from hypergrep_amd import device

    n = len(data)
    buf = torch.zeros(n + 32, dtype=torch.uint8, device="cuda:0")
    if n:
        buf[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    db = device.Database(patterns, flags=flags, ids=ids)
    sc = device.Scanner(db, 0)
    stats = sc.scan(buf.data_ptr(), n, buffer_size=buffer_size, line_base=line_base)
    return sorted(sc.hits()), stats


def oracle_hits(data, patterns, flags=None, ids=None, buffer_size=262140):
    rc, hits, nlines = oracle_py.scan_buffer(data, patterns, flags=flags, ids=ids, buffer_size=buffer_size)
    assert rc == 0
    return sorted(hits), nlines


# ------------------------------------------------------------------ Face B: hyperscan() through the Python API mirror
@pytest.mark.parametrize("v", [v for v in VECTORS if v["data"] is not None], ids=lambda v: v["name"])
def test_plumbing_vectors_face_b(torch_cuda, v, tmp_path):
    import hypergrep_amd

    data = base64.b64decode(v["data"])
    path = tmp_path / "in.txt"
    path.write_bytes(data)
    rows, batches = [], []

    def cb(matches, count):
        batches.append(count)
        for i in range(count):
            rows.append((matches[i].line_number, matches[i].id, matches[i].line))

    rc = hypergrep_amd.scan(str(path), v["patterns"], cb, **v["kwargs"])
    assert _loaded_native()
    assert rc == v["rc"]
    assert rows == [(r[0], r[1], base64.b64decode(r[2])) for r in v["rows"]]
    assert batches == v["batches"]


def test_missing_file_rc6(torch_cuda):
    import hypergrep_amd

    called = []
    rc = hypergrep_amd.scan("/nonexistent/definitely/missing", ["x"], lambda m, c: called.append(c))
    assert rc == 6 and not called


@pytest.mark.parametrize("case", TABLES["scan"], ids=lambda c: c["name"])
def test_reference_table_scan(torch_cuda, case, capsys):
    import hypergrep_amd

    def cb(matches, count):  # the reference test's _basic_callback (test_hypergrep.py:18-23)
        for i in range(count):
            print(f"{matches[i].line_number}:{matches[i].line.decode(errors='ignore').rstrip()}")

    hypergrep_amd.scan(os.path.join(FILES, case["file"]), case["patterns"], cb)
    assert capsys.readouterr().out.splitlines() == case["returns"]


@pytest.mark.parametrize("case", TABLES["grep"], ids=lambda c: c["name"])
def test_reference_table_grep(torch_cuda, case):
    import hypergrep_amd

    path = FILES if case["file"] == "." else os.path.join(FILES, case["file"])
    if "raises" in case:
        exc = {"FileNotFoundError": FileNotFoundError, "ValueError": ValueError}[case["raises"]]
        with pytest.raises(exc):
            hypergrep_amd.grep(path, case["patterns"], **case["kwargs"])
    else:
        got = hypergrep_amd.grep(path, case["patterns"], **case["kwargs"])
        assert [[list(t) for t in got[0]], got[1]] == case["returns"]


@pytest.mark.parametrize("case", TABLES["check_compatibility"], ids=lambda c: c["name"])
def test_reference_table_check(case):
    import hypergrep_amd

    assert hypergrep_amd.check_compatibility(case["patterns"]) == case["returns"]


def test_grep_counts_on_greptest_files(torch_cuda):
    import hypergrep_amd

    # expectations from the reference's parallel_grep table (test_hypergrep.py:324-418, 553-687)
    f1 = os.path.join(FILES, "greptest1.txt")
    assert hypergrep_amd.grep(f1, ["foo"], count_only=True) == (16, 0)
    assert hypergrep_amd.grep(f1, ["foo"], max_match_count=3) == ([(2, "foo\n"), (3, "foobar\n"), (4, "[foo]\n")], 0)
    assert hypergrep_amd.grep(f1, ["fOoBaR"]) == ([], 0)
    assert hypergrep_amd.grep(f1, ["fOoBaR"], ignore_case=True) == ([(3, "foobar\n")], 0)
    assert hypergrep_amd.grep(f1, ["barfoo\\+"]) == ([(12, "barfoo+\n")], 0)
    assert hypergrep_amd.grep(f1, ["barfoo+"]) == ([(11, "barfoo\n"), (12, "barfoo+\n")], 0)
    assert hypergrep_amd.grep(f1, ["foobar", "fo{2}bar", "fo+bar"]) == ([(3, "foobar\n")], 0)
    assert hypergrep_amd.grep(f1, ["foobar", "extra foo bar"]) == ([(3, "foobar\n"), (16, "extra foo bar\n")], 0)
    got = hypergrep_amd.grep(f1, ["grep file to test|sync with"], only_matching=True)
    assert got == ([(1, "grep file to test\n"), (1, "sync with\n"), (18, "grep file to test\n"), (18, "sync with\n")], 0)


# ------------------------------------------------------------------ hg_* device-buffer API vs the oracle
def _log_text(rng, nlines, needles, p_hit=0.2, maxlen=160):
    words = ["alpha", "beta", "gamma", "delta", "status=200", "user=bob", "GET", "/index.html", "10.0.0.1", "ok",
             "warn", "retry", "timeout=30", "id=12345", "x", "user=guest status=404"]
    out = []
    for _ in range(nlines):
        n = rng.randint(0, 14)
        toks = [rng.choice(words) for _ in range(n)]
        if needles and rng.random() < p_hit:
            toks.insert(rng.randint(0, len(toks)), rng.choice(needles))
        out.append(" ".join(toks)[:maxlen])
    return ("\n".join(out) + "\n").encode()


MIXED_PATTERNS = ["needle_in_haystack", "ERR_DISK_FULL_[0-9]{3}", "user=[a-z0-9_]{4,12} status=5[0-9]{2}",
                  "(?i)caseless_needle", "(first_long_alt|second_long_alt) tail", "connection reset by peer$",
                  "^kernel panic -", "\\bwordbound_token\\b", "warn", "x$", "[0-9]+\\.[0-9]+\\.[0-9]+"]
MIXED_NEEDLES = ["needle_in_haystack", "ERR_DISK_FULL_042", "ERR_DISK_FULL_04", "user=alice_01 status=503",
                 "user=al status=503", "CaseLess_Needle", "first_long_alt tail", "second_long_alt tail",
                 "second_long_alt  tail", "connection reset by peer", "kernel panic - not syncing", "wordbound_token",
                 "xwordbound_tokenx", "needle_in_haystac"]


@pytest.mark.parametrize("ids_mode", ["shared", "distinct"])
def test_mixed_tiers_match_oracle(torch_cuda, ids_mode):
    rng = random.Random(77)
    data = _log_text(rng, 30000, MIXED_NEEDLES)
    ids = None if ids_mode == "shared" else list(range(len(MIXED_PATTERNS)))
    want, nlines = oracle_hits(data, MIXED_PATTERNS, ids=ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, MIXED_PATTERNS, ids=ids)
    assert stats.n_lines == nlines
    assert got == want
    assert len(want) > 1000


def test_tile_boundaries_long_lines_and_buffer_sizes(torch_cuda):
    rng = random.Random(3)
    pat = ["needle_in_haystack", "tail_anchor_zz$"]
    chunks = []
    for _ in range(60):
        chunks.append(b"a" * rng.randint(0, 9000) + b" needle_in_haystack " + b"b" * rng.randint(0, 9000) + b" tail_anchor_zz\n")
    chunks.append(b"q" * 70000 + b"needle_in_haystack" + b"r" * 70000 + b"tail_anchor_zz\n")
    chunks.append(b"short needle_in_haystack\n")
    chunks.append(b"no newline at end needle_in_haystack tail_anchor_zz")
    data = b"".join(chunks)
    for bs in (262140, 20001, 16385, 16384, 4097, 100, 9):
        want, nlines = oracle_hits(data, pat, ids=[0, 1], buffer_size=bs)
        got, stats = gpu_scan_buffer(torch_cuda, data, pat, ids=[0, 1], buffer_size=bs)
        assert stats.n_lines == nlines, bs
        assert got == want, bs


def test_straddle_every_alignment(torch_cuda):
    lit = "needle_in_haystack"
    for shift in range(0, 40, 3):
        data = b"x" * (16384 - 20 + shift) + lit.encode() + b"\nnext line\n"
        want, _ = oracle_hits(data, [lit])
        got, _ = gpu_scan_buffer(torch_cuda, data, [lit])
        assert got == want and len(got) == 1


def test_nul_rules_and_edge_inputs(torch_cuda):
    pats = ["needle_in_haystack", "x"]
    data = (b"\0\0needle_in_haystack\n" b"ab\0needle_in_haystack x\n" b"x\0\0\n" b"\0\n" b"needle_in_haystack\0x\n" b"\0\0\0")
    for ids in (None, [1, 2]):
        want, nlines = oracle_hits(data, pats, ids=ids)
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, ids=ids)
        assert got == want and stats.n_lines == nlines
    for data in (b"", b"\n", b"x", b"\n\n\n", b"needle_in_haystack"):
        want, nlines = oracle_hits(data, pats, ids=[1, 2])
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, ids=[1, 2])
        assert got == want and stats.n_lines == nlines


ALWAYS_ON_POOL = [
    "[0-9]+\\.[0-9]+", "\\b[xyz]{2}\\b", " +[a-c]", "[a-c]+=[0-9]", "x.y", "[^a]b", "a.", "\\Bab", "^[a-c]", "[0-9]$", "[a-c]{2,5}x",
    "(?:ab|c)+z", "\\b[0-9]{3}\\b", "_[a-z]*-", "[xyz]+\\b", "=\\B", "[a-c][0-9][a-c]", "y[^\\n]*z", "\\s[xyz]", "[[:digit:]]+[a-c]", "(?i)xY", "0*1",
    "[0-9]+\\s", "x[^y]", "[a-c]+\\W", "\\b[xyz]+\\s",
    # two state words (33..64 positions)
    "[^ ]{34}", "[^ =]{36}x?", "[a-z0-9_.-]{33,}", "\\b[^ ]{33}", "[^ ]{20}[a-c][^ ]{20}",
    "(?:ab|c)[^ ]{35}", "[^ ]{16}x?[^ ]{20}", "[^ ]{12}(?:[a-c][0-9])*[^ ]{24}", "[^ ]{30}[^ ]?[^ ]?[^ ]?y",
]


@pytest.mark.parametrize("seed", range(26))
def test_always_on_tier_multi_tile(torch_cuda, seed):
    """Expressions without a usable required literal over texts of many tiles (hg_always_on_fast_kernel): lean dword steps in
    the tiles inside the text, the exact per-byte routine in the last tile, for expressions whose match can include the newline
    ("a.", "\\s[xyz]") and for small scan buffers; groups of several expressions in one state word with and without boundary
    conditions; SINGLEMATCH and all-matches flags; lines longer than the lanes' segments and than the scan buffer; NULs."""
    rng = random.Random(777000 + seed)
    k = rng.choice([1, 2, 3, 4, 6])
    pats = [rng.choice(ALWAYS_ON_POOL) for _ in range(k)]
    flags = [rng.choice([14, 14, 6, 10, 2, 15]) for _ in pats]
    ids = [rng.randint(0, 2) for _ in pats] if rng.random() < 0.5 else list(range(k))
    assert oracle_py.check_patterns(pats, flags=flags) == 0
    maxlen = rng.choice([24, 200])
    data = bytearray(regex_gen.random_text(rng, rng.choice([1500, 4000]) * (6 if maxlen == 24 else 1), maxlen=maxlen, final_newline=rng.random() < 0.8))
    if seed % 2:
        at = rng.randrange(len(data))
        data[at:at] = bytes(rng.choice(b"abcxyz01 ._-=") for _ in range(rng.choice([700, 5000, 40000])))  # a line longer than a segment
        for _ in range(rng.randint(0, 6)):
            data[rng.randrange(len(data))] = 0
    data = bytes(data)
    assert len(data) > 3 * 16384
    for bs in ((262140, 1000) if seed % 4 == 1 else (262140,)):
        want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=bs)
        assert got == want, (pats, flags, ids, bs, len(got), len(want), sorted(set(got) - set(want))[:3], sorted(set(want) - set(got))[:3])
        assert stats.n_lines == nlines
    assert _loaded_native()


def test_not_singlematch_and_mixed_ids(torch_cuda):
    data = b"aaa needle_in_haystack needle_in_haystack\nba\n"
    pats, flags, ids = ["a", "needle_in_haystack", "needle"], [6, 6, 14], [0, 1, 1]
    want, _ = oracle_hits(data, pats, flags, ids)
    got, _ = gpu_scan_buffer(torch_cuda, data, pats, flags, ids)
    assert got == want


@pytest.mark.parametrize("seed", range(6))
def test_random_patterns_match_oracle(torch_cuda, seed):
    rng = random.Random(9000 + seed)
    done = 0
    for _ in range(12):
        k = rng.randint(1, 4)
        pats = [regex_gen.random_pattern(rng) for _ in range(k)]
        flags = [rng.choice([14, 14, 15, 10, 6, 12]) for _ in range(k)]
        ids = [rng.randint(0, 2) for _ in range(k)]
        if oracle_py.check_patterns(pats, flags=flags) != 0:
            continue
        data = regex_gen.random_text(rng, 400, final_newline=rng.random() < 0.8)
        want, nlines = oracle_hits(data, pats, flags, ids)
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids)
        assert got == want, (pats, flags, ids)
        assert stats.n_lines == nlines
        done += 1
    assert done >= 4


def test_many_literals_dense_hits_and_workspace_growth(torch_cuda):
    rng = random.Random(11)
    lits = ["tok_%04x_%s" % (i, "".join(rng.choice("abcdef") for _ in range(rng.randint(2, 10)))) for i in range(600)]
    data = _log_text(rng, 60000, lits, p_hit=0.9)  # ~0.9 hits per line: overflows the initial workspace
    ids = list(range(len(lits)))
    want, nlines = oracle_hits(data, lits, ids=ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, lits, ids=ids)
    assert stats.n_lines == nlines and got == want
    assert len(want) > 40000


def test_large_literal_set_wide_filter(torch_cuda):
    """BASELINE config 5 (4096 literals, 16384 windows): the wide LDS filter (two 16-bit fingerprints per slot)."""
    from hypergrep_amd import benchspec, device

    pats, needles, hpm = benchspec.c5_spec()
    ids = list(range(len(pats)))
    data = device.synth_host(192 << 10, benchspec.SEED_BASE + 5, needles, hpm)
    want, nlines = oracle_hits(data, pats, ids=ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, pats, ids=ids)
    assert stats.n_lines == nlines and got == want and len(want) > 100
    # and at a size where every tile path runs (count cross-checked against the generator's own hit rate)
    torch = torch_cuda
    nbytes = 256 << 20
    text = torch.empty(nbytes + 32, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, seed=benchspec.SEED_BASE + 5, needles=needles, hit_per_million=hpm)
    sc = device.Scanner(device.Database(pats, ids=ids), 0)
    st = sc.scan(text.data_ptr(), nbytes)
    head = sorted(h[:3] for h in sc.hits(limit=len(want) + 64) if h[0] < nlines - 1)  # the prefix's last line is cut short
    assert head == [w[:3] for w in want if w[0] < nlines - 1]
    assert abs(st.n_hits / st.n_lines - hpm / 1e6) < 0.02 * hpm / 1e6 + 1e-4


def test_synthetic_log_device_equals_host_and_oracle(torch_cuda):
    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    patterns, needles, hpm = benchspec.c3_spec()
    nbytes = (8 << 20) + 12345  # not a multiple of the generator block nor of the scan tile
    text = torch.empty(nbytes + 32, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, seed=42, needles=needles, hit_per_million=hpm * 4, first_block=5)
    host = bytes(text[:nbytes].cpu().numpy())
    assert host == device.synth_host(nbytes, 42, needles, hpm * 4, first_block=5)
    for ids in (None, list(range(len(patterns)))):
        db = device.Database(patterns, ids=ids)
        sc = device.Scanner(db, 0)
        stats = sc.scan(text.data_ptr(), nbytes, line_base=1000)
        want, nlines = oracle_hits(host, patterns, ids=ids)
        assert stats.n_lines == nlines
        assert sorted(sc.hits()) == sorted((ln + 1000, i, to, off, ln_len) for ln, i, to, off, ln_len in want)


def test_large_file_chunked_face_b(torch_cuda, tmp_path, monkeypatch):
    """hyperscan() with a chunk size far below the file size: chunk cuts, carried line numbers, batching."""
    import hypergrep_amd
    from hypergrep_amd import benchspec, device

    patterns, needles, hpm = benchspec.c3_spec()
    nbytes = 5 << 20
    host = device.synth_host(nbytes, 9, needles, hpm * 3)
    host = host[: nbytes - 777]  # last line without '\n'
    path = tmp_path / "log.txt"
    path.write_bytes(host)
    monkeypatch.setenv("HYPERGREP_CHUNK_MB", "1")
    rows = []

    def cb(matches, count):
        for i in range(count):
            rows.append((matches[i].line_number, matches[i].id, matches[i].line))

    ids = list(range(len(patterns)))
    rc = hypergrep_amd.scan(str(path), patterns, cb, ids=ids, buffer_count=64)
    assert rc == 0
    orc, want, _ = oracle_py.scan_file(str(path), patterns, ids=ids, buffer_count=64)
    assert orc == 0 and rows == want and len(rows) > 500


# ------------------------------------------------------------------ Face A: the six libhs symbols, block mode
class _HsErr(__import__("ctypes").Structure):
    _fields_ = [("message", __import__("ctypes").c_char_p), ("expression", __import__("ctypes").c_int)]


def _hs_events(lib, patterns, flags, ids, blocks):
    """Compile + scan each block through the libhs face of `lib`; returns [[(id, to), ...] per block]."""
    import ctypes

    handler_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_uint, ctypes.c_void_p)
    n = len(patterns)
    pa = (ctypes.c_char_p * n)(*[p.encode() for p in patterns])
    fa = (ctypes.c_uint * n)(*flags)
    ia = (ctypes.c_uint * n)(*ids)
    db, err, scratch = ctypes.c_void_p(), ctypes.POINTER(_HsErr)(), ctypes.c_void_p()
    lib.hs_compile_multi.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.c_uint,
                                     ctypes.c_uint, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.POINTER(_HsErr))]
    lib.hs_alloc_scratch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.hs_scan.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, handler_t, ctypes.c_void_p]
    lib.hs_free_scratch.argtypes = [ctypes.c_void_p]
    lib.hs_free_database.argtypes = [ctypes.c_void_p]
    lib.hs_free_compile_error.argtypes = [ctypes.c_void_p]
    rc = lib.hs_compile_multi(pa, fa, ia, n, 1, None, ctypes.byref(db), ctypes.byref(err))
    lib.hs_free_compile_error(err)
    assert rc == 0
    assert lib.hs_alloc_scratch(db, ctypes.byref(scratch)) == 0
    out = []
    for block in blocks:
        events = []

        def on_event(id_, from_, to, flags_, ctx, events=events):
            events.append((id_, to))
            return 0

        assert lib.hs_scan(db, block, len(block), 0, scratch, handler_t(on_event), None) == 0
        out.append(events)
    lib.hs_free_scratch(scratch)
    lib.hs_free_database(db)
    return out


def test_face_a_hs_scan_matches_oracle(torch_cuda):
    import ctypes

    from hypergrep_amd import utils

    product = utils._get_hyperscanner_lib()
    oracle = ctypes.CDLL(os.path.join(oracle_py.ORACLE_DIR, "_build", "libhs.so.5"))
    patterns = ["needle_in_haystack", "fo+bar", "^begin", "end$", "a.c", "\\bword\\b", "(?i)CaseLess_Long_Literal", "x"]
    flags = [14, 14, 14, 14, 6, 10, 14, 6]
    ids = [0, 1, 2, 3, 4, 5, 6, 7]
    blocks = [b"foobar\n", b"begin with needle_in_haystack and end\n", b"two\nbegin lines end\nneedle_in_haystack", b"a\nc abc word\n",
              b"xx caseless_long_literal CASELESS_LONG_LITERAL\n", b"\0x\0needle_in_haystack\n", b"q" * 40000 + b"needle_in_haystack" + b"z" * 100]
    got = _hs_events(product, patterns, flags, ids, blocks)
    want = _hs_events(oracle, patterns, flags, ids, blocks)
    assert got == want
    assert any(len(e) > 2 for e in want)
    # early termination: a handler that returns non-zero stops the scan with HS_SCAN_TERMINATED (-3)
    handler_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_uint, ctypes.c_void_p)
    n = 1
    pa = (ctypes.c_char_p * n)(b"a")
    fa = (ctypes.c_uint * n)(6)
    ia = (ctypes.c_uint * n)(0)
    db, scratch = ctypes.c_void_p(), ctypes.c_void_p()
    assert product.hs_compile_multi(pa, fa, ia, n, 1, None, ctypes.byref(db), None) == 0
    assert product.hs_alloc_scratch(db, ctypes.byref(scratch)) == 0
    seen = []
    assert product.hs_scan(db, b"aaaa", 4, 0, scratch, handler_t(lambda i, f, t, fl, c: seen.append(t) or 1), None) == -3
    assert seen == [1]
    product.hs_free_scratch(scratch)
    product.hs_free_database(db)
    bad = ctypes.POINTER(_HsErr)()
    assert product.hs_compile_multi((ctypes.c_char_p * 1)(b"(?<!a)b"), fa, ia, 1, 1, None, ctypes.byref(db), ctypes.byref(bad)) == -4
    assert bad.contents.expression == 0 and bad.contents.message
    product.hs_free_compile_error(bad)


def test_face_a_short_block_regimes(torch_cuda):
    """hs_scan has three regimes by block length: split over lanes by start position (<= 2047 bytes), one lane per
    expression (<= 8192), the general pipeline above; and the short-block kernel stages the automaton tables in LDS only when
    they fit (<= 2048 expressions, 32 per workgroup).  Lengths on both sides of each boundary, expressions whose state
    never dies (.*), multi-word automata, SINGLEMATCH and all-matches expressions sharing ids, and a 2100-expression set."""
    import ctypes
    import random

    from hypergrep_amd import utils

    product = utils._get_hyperscanner_lib()
    oracle = ctypes.CDLL(os.path.join(oracle_py.ORACLE_DIR, "_build", "libhs.so.5"))
    rng = random.Random(77)
    patterns = ["needle_in_haystack", "fo+bar[0-9]*", "a.c", "st.*us=2", "^\\S+ \\S+", "[a-z]{3,40}=[0-9]{2,30} [a-z]{10,60}x", "\\bGET\\b", "0$", "(?i)error.*timeout"]
    flags = [14, 6, 6, 6, 14, 6, 10, 14, 6]
    ids = [0, 1, 2, 2, 3, 4, 5, 6, 1]
    filler = regex_gen.random_text(rng, 400, maxlen=120) + b"status=200 GET /a.c needle_in_haystack foobar12 ERROR x timeout\n"
    blocks = []
    for n in (1, 7, 8, 9, 63, 64, 65, 511, 2040, 2046, 2047, 2048, 2049, 4095, 8191, 8192, 8193, 20000):
        at = rng.randrange(0, len(filler) - n) if n < len(filler) else 0
        blocks.append(filler[at:at + n] if n < len(filler) else (filler * 2)[:n])
        blocks.append(filler[-n:])  # ends with the dense line and its newline
    got = _hs_events(product, patterns, flags, ids, blocks)
    want = _hs_events(oracle, patterns, flags, ids, blocks)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, (k, len(blocks[k]), sorted(set(g) - set(w))[:4], sorted(set(w) - set(g))[:4])
    assert sum(len(w) for w in want) > 500
    # a set beyond 64 workgroups of 32: 256 expressions per workgroup, tables read from HBM
    many = [f"lit{i:05d}x" for i in range(2100)] + ["status=[0-9]+", "a.c"]
    mflags = [14] * 2100 + [6, 6]
    mids = [i % 50 for i in range(2100)] + [50, 51]
    mblocks = [b"lit00007x lit02099x status=200 abc\n", b"nothing here\n", filler[:2047], filler[:3000] + b"lit01234x\n"]
    assert _hs_events(product, many, mflags, mids, mblocks) == _hs_events(oracle, many, mflags, mids, mblocks)


def test_chunked_pipeline_matches_oracle(torch_cuda, monkeypatch):
    """Force the two-stream chunked pipeline (normally used above 512 MiB) on a 40 MiB text: chunk-crossing lines, carried
    line numbers and double-buffered candidate segments must give the same hits as the oracle."""
    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    patterns, needles, hpm = benchspec.c3_spec(n_literals=12, n_classes=8, n_anchored=8)
    nbytes = (40 << 20) + 4321
    text = torch.empty(nbytes + 32, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, seed=1234, needles=needles, hit_per_million=hpm * 3)
    host = bytes(text[:nbytes].cpu().numpy())
    ids = list(range(len(patterns)))
    want, nlines = oracle_hits(host, patterns, ids=ids)
    monkeypatch.setenv("HG_CHUNK_TILES", "1024")  # 16 MiB chunks -> 3 chunks
    db = device.Database(patterns, ids=ids)
    db.tune(host[: 1 << 20])
    sc = device.Scanner(db, 0)
    stats = sc.scan(text.data_ptr(), nbytes)
    assert stats.n_lines == nlines
    assert sorted(sc.hits()) == want and len(want) > 1000
    # the same text in two segments of two pipeline chunks each (a pass may hold fewer reports than the text has)
    monkeypatch.setenv("HG_HIT_LIMIT", str(int(stats.n_raw_hits * 0.6)))
    sc_seg = device.Scanner(db, 0)
    st_seg = sc_seg.scan(text.data_ptr(), nbytes)
    assert st_seg.n_lines == nlines and st_seg.stream_launches >= 4
    assert sc_seg.hits() == want
    monkeypatch.delenv("HG_HIT_LIMIT")
    # ... and when the text needs more pipeline chunks than a pass has (64; lowered here to 2)
    monkeypatch.setenv("HG_MAX_CHUNKS", "2")
    sc_seg2 = device.Scanner(db, 0)
    st_seg2 = sc_seg2.scan(text.data_ptr(), nbytes)
    assert st_seg2.n_lines == nlines and st_seg2.stream_launches >= 4
    assert sc_seg2.hits() == want
    monkeypatch.delenv("HG_MAX_CHUNKS")
    # and again with a pattern from the always-on tier in the mix
    pats2 = patterns + ["warn|retry"]
    want2, _ = oracle_hits(host[: 20 << 20], pats2, ids=ids + [999])
    sc2 = device.Scanner(device.Database(pats2, ids=ids + [999]), 0)
    sc2.scan(text.data_ptr(), 20 << 20)
    assert sorted(sc2.hits()) == want2


def test_full_size_properties_32gib(torch_cuda, monkeypatch):
    """BASELINE config 3 at its full size (256 patterns, 32 GiB in HBM): size-independent properties instead of an oracle
    run — exact line count, strict (line, id, to) order, idempotence, the chunked pipeline and the single-pass path agree
    record for record, and the hits of the first 2 MiB equal the oracle's."""
    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < 80 << 30:
        pytest.skip("needs 80 GiB of free HBM")
    patterns, needles, hpm = benchspec.c3_spec()
    ids = list(range(len(patterns)))
    nbytes = 32 << 30
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + 3, needles, hpm)
    torch.cuda.synchronize()
    newlines = sum(int((text[o:min(o + (4 << 30), nbytes)] == 10).sum()) for o in range(0, nbytes, 4 << 30))
    db = device.Database(patterns, ids=ids)
    # tuned as the file API tunes by itself (hg_shim.hip maybe_tune) and as bench.py does: four pieces of 256 KiB spread over the
    # first 256 MiB
    db.tune(b"".join(bytes(text[(i * (64 << 20)) & ~15: ((i * (64 << 20)) & ~15) + (256 << 10)].cpu().numpy()) for i in range(4)))
    scanners = [device.Scanner(db, 0)]

    def run():
        sc = scanners[0]
        st = sc.scan(text.data_ptr(), nbytes)
        buf = torch.empty((st.n_hits, 2), dtype=torch.int64, device="cuda:0")
        assert sc.copy_hits_to(buf.data_ptr(), st.n_hits) == st.n_hits
        torch.cuda.synchronize()
        return st, buf

    st, a = run()
    assert st.n_lines == newlines + (0 if int(text[nbytes - 1]) == 10 else 1)  # every line is one piece; the last one may be cut short
    assert st.stream_launches == 4  # the chunked pipeline
    # record = (u64 line, u32 id, u32 to): strictly increasing in (line, id, to)
    line, rest = a[:, 0], a[:, 1]
    ident, to = rest & 0xFFFFFFFF, (rest >> 32) & 0xFFFFFFFF
    key = (ident << 32) | to
    same_line = line[1:] == line[:-1]
    assert bool(((line[1:] > line[:-1]) | (same_line & (key[1:] > key[:-1]))).all())
    assert 0.009 < st.n_hits / st.n_lines < 0.012  # 1.05 % of lines carry a matching needle
    st2, b = run()
    assert st2.n_hits == st.n_hits and bool((a == b).all())  # idempotent
    monkeypatch.setenv("HG_CHUNK_TILES", str(1 << 30))  # one pass, no side stream
    scanners[0] = device.Scanner(db, 0)  # (a scanner reads the engine's knobs when it is created)
    st3, c = run()
    assert st3.stream_launches == 1 and st3.n_hits == st.n_hits and st3.n_lines == st.n_lines and bool((a == c).all())
    monkeypatch.delenv("HG_CHUNK_TILES")
    # the head of the text against the oracle
    head_n = 2 << 20
    host = bytes(text[:head_n].cpu().numpy())
    head_n = host.rfind(b"\n") + 1
    want, nl = oracle_hits(host[:head_n], patterns, ids=ids)
    got = [h[:3] for h in scanners[0].hits(limit=len(want) + 16) if h[0] < nl]
    assert sorted(got) == [w[:3] for w in want]


def test_crowded_filter_slots_and_large_buckets(torch_cuda):
    """Stress of the paths a small pattern set never reaches: (a) 3000 literals -> slots shared by three and more windows
    (HgSlotInfo.many, weak care masks); (b) 200 automaton patterns behind ONE shared literal, dense in the text -> buckets of
    200 pairs per candidate, the verify pass's LDS stage overflows into its direct path, pattern-keyed lists fill unevenly."""
    rng = random.Random(2024)
    alphabet = "abcdefghijklmnopqrstuvwxyz0123456789_"
    lits = sorted({"".join(rng.choice(alphabet) for _ in range(rng.randint(8, 14))) for _ in range(3000)})
    data = _log_text(rng, 8000, lits, p_hit=0.5)
    ids = list(range(len(lits)))
    want, nlines = oracle_hits(data, lits, ids=ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, lits, ids=ids)
    assert stats.n_lines == nlines and got == want and len(want) > 3000

    pats = [f"shared_literal_x{i % 10}[0-9]{{1,3}}(?:a{i}b|c{i}d)" for i in range(200)]
    needles = [f"shared_literal_x{i % 10}{rng.randint(0, 999)}{'a%db' % i if i % 2 else 'c%dd' % i}" for i in range(200)] + ["shared_literal_x3", "shared_literal_x77zz"]
    data = _log_text(rng, 20000, needles, p_hit=0.95)
    ids = list(range(len(pats)))
    want, nlines = oracle_hits(data, pats, ids=ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, pats, ids=ids)
    assert stats.n_lines == nlines and got == want and len(want) > 10000


@pytest.mark.parametrize("seed", range(8))
def test_random_literal_anchored_patterns_match_oracle(torch_cuda, seed):
    """Random expressions around a long literal: every one goes through the prefilter, the verify pass and one of the
    confirm routines (literal-only, one-word, two-word / boundary, generic for all-matches mode)."""
    rng = random.Random(7300 + seed)
    pairs = [regex_gen.anchored_pattern(rng) for _ in range(rng.randint(3, 24))]
    pats = [p for p, _ in pairs]
    flags = [rng.choice([14, 14, 15, 6, 10]) for _ in pats]
    ids = [rng.randint(0, 5) for _ in pats]
    assert oracle_py.check_patterns(pats, flags=flags) == 0, pats
    data = regex_gen.anchored_text(rng, [s for _, s in pairs], 6000)
    want, nlines = oracle_hits(data, pats, flags, ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids)
    assert stats.n_lines == nlines
    assert got == want, (pats, flags, ids)
    assert len(want) > 500


@pytest.mark.parametrize("mib, chunk_tiles, seed, extra", [(97, 1024, 1, []), (333, 4096, 2, []), (700, 8192, 3, []), (1311, 16384, 4, []),
                                                            (211, 2048, 5, ["retry", "=77", "=7[0-9]? "])])
def test_chunked_pipeline_equals_single_pass(torch_cuda, monkeypatch, mib, chunk_tiles, seed, extra):
    """The chunked two-stream pipeline (early sort of the first chunks + merge, carried line numbers, double-buffered
    candidates) and the single pass must deliver identical records, for chunk counts from 3 to 6 and ragged sizes."""
    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    patterns, needles, hpm = benchspec.c3_spec()
    patterns = patterns + extra  # short literals: byte-aligned probing plus an always-on expression in the chunked pipeline
    ids = list(range(len(patterns)))
    nbytes = (mib << 20) + 12345 * seed
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, 5000 + seed, needles, hpm * 2)
    torch.cuda.synchronize()
    db = device.Database(patterns, ids=ids)
    assert db.info()["byte_windows"] == (1 if extra else 0)

    def run():
        sc = device.Scanner(db, 0)  # (a scanner reads the engine's knobs — HG_CHUNK_TILES below — when it is created)
        st = sc.scan(text.data_ptr(), nbytes, line_base=77)
        buf = torch.empty((st.n_hits, 2), dtype=torch.int64, device="cuda:0")
        sc.copy_hits_to(buf.data_ptr(), st.n_hits)
        torch.cuda.synchronize()
        return st, buf

    monkeypatch.setenv("HG_CHUNK_TILES", str(1 << 30))
    one, a = run()
    assert one.stream_launches == 1
    monkeypatch.setenv("HG_CHUNK_TILES", str(chunk_tiles))
    many, b = run()
    assert many.stream_launches >= 3
    assert (many.n_hits, many.n_lines) == (one.n_hits, one.n_lines) and one.n_hits > 10000
    assert bool((a == b).all())


def test_always_on_patterns_segment_scan(torch_cuda):
    """Patterns without a long required literal (always-on tier): bounded ones scan 256-byte segments with max_len - 1 bytes
    of lead-in, unbounded ones from their line's start; tiles with forced breaks or a long carry-in line go to the scalar
    routine.  Text with lines from empty to 40 KiB, NULs, CRLF, and matches across segment and tile borders."""
    rng = random.Random(99)
    pats = ["ab", "x", "foo|bar", "[0-9]+\\.[0-9]+", "a.*z", "\\bGET\\b", "^warn", "end$", "(?i)eRRoR", "q{2,3}", "[^a-z \\n]{3}"]
    flags = [14, 14, 14, 14, 14, 14, 14, 14, 15, 6, 6]  # the last two report every match end (no SINGLEMATCH)
    ids = list(range(len(pats)))
    words = ["ab", "x", "foo", "bar", "12.5", "7.", "a", "z", "GET", "GETS", "warn", "end", "Error", "ERROR", "qq", "qqqq", "A1B2", "", "lorem", "ipsum"]
    lines = []
    for i in range(6000):
        n = rng.choice([0, 1, 3, 8, 20, 20, 60, 300]) if i % 500 else 9000  # every 500th line is ~40 KiB: longer than two tiles
        toks = [rng.choice(words) for _ in range(n)]
        line = " ".join(toks)
        if i % 97 == 0:
            line = line[: len(line) // 2] + "\0" + line[len(line) // 2:]
        if i % 131 == 0:
            line = "\0\0" + line
        if i % 53 == 0:
            line += "\r"
        lines.append(line)
    data = ("\n".join(lines) + ("\n" if rng.random() < 0.5 else "")).encode()
    want, nlines = oracle_hits(data, pats, flags, ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids)
    assert stats.n_lines == nlines
    assert got == want and len(want) > 20000
    # forced breaks: a scan buffer shorter than the long lines
    want2, nlines2 = oracle_hits(data, pats, flags, ids, buffer_size=20000)
    got2, stats2 = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=20000)
    assert stats2.n_lines == nlines2 and got2 == want2


@pytest.mark.gpu
def test_short_literals_byte_aligned_windows(torch_cuda):
    """Required literals of 3..6 bytes switch the prefilter to byte-aligned probing (one window per literal, 16 probes per
    16 bytes).  Occurrences at every alignment, across 1 KiB rows, tile edges and the end of the text, NULs, forced breaks."""
    from hypergrep_amd import device

    pats = ["ERROR", "WARN", "foo", "(?i)Fail", "a\\.b", "panic: [a-z]+", "x=\\d+;", "status=5[0-9]{2}", "\\bGET\\b /api"]
    flags = [14, 14, 10, 14, 14, 6, 14, 14, 14]
    ids = [0, 1, 2, 3, 4, 5, 6, 7, 7]
    info = device.Database(pats, flags, ids).info()
    assert info["byte_windows"] == 1 and info["n_always_on"] == 1
    rng = random.Random(78)
    words = [b"ERROR", b"WARN", b"foo", b"FAIL", b"fAiL", b"a.b", b"panic: oops", b"x=12;", b"ERRO", b"WAR", b"fo", b"fai", b"x=;", b"foofoo",
             b"status=503", b"status=200", b"GET /api", b"GETS /api"]
    for trial in range(3):
        out = bytearray()
        while len(out) < 300000:
            line = bytearray()
            for _ in range(rng.choice([0, 2, 6, 6, 40])):
                line += rng.choice(words) if rng.random() < 0.4 else bytes(rng.choice(b"abcdefoOrRE =.;0123") for _ in range(rng.randint(1, 9)))
                if rng.random() < 0.5:
                    line += b" "
            if rng.random() < 0.02:
                line[len(line) // 2:len(line) // 2] = b"\0"
            out += line + b"\n"
        out[100000:100000] = b"z" * 30000 + b" ERROR foo " + b"y" * 20000  # one line longer than a tile
        for at in (1021, 1022, 1023, 2045, 16381, 16382, 16383, 32765, 65533):
            out[at:at + 5] = b"ERROR"
            out[at + 3000:at + 3003] = b"foo"
        data = bytes(out[:290000 + trial]) + rng.choice([b"foo", b"WARN", b"ERROR\n", b"fo"])
        for bs in (262140, 4096):
            want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
            got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=bs)
            assert stats.n_lines == nlines and got == want, (trial, bs)
        assert len(want) > 3000


@pytest.mark.gpu
def test_many_three_byte_literals_three_byte_windows(torch_cuda):
    """400 three-byte literals: too many to enumerate the byte after each, so the whole set uses 3-byte windows (hash
    weights with a zero top byte) under byte-aligned probing; nothing may fall back to the always-on tier."""
    from hypergrep_amd import device

    rng = random.Random(32)
    alphabet = "abcdefghijklmnopqrstuvwxyz"
    words = sorted({"".join(rng.choice(alphabet) for _ in range(3)) for _ in range(400)})
    pats = words + ["needle", "(?i)MiXeD", "ab[0-9]x", "tail$"]
    flags = [14] * len(words) + [14, 14, 6, 14]
    ids = list(range(len(pats)))
    info = device.Database(pats, flags, ids).info()
    assert info["byte_windows"] == 1 and info["n_always_on"] == 0 and info["n_windows"] == info["n_factors"]
    out = bytearray()
    while len(out) < 400000:
        toks = [rng.choice(words + ["needle", "mixed", "MIXED", "ab7x", "tail", "zz", "q"]) if rng.random() < 0.3
                else "".join(rng.choice(alphabet + "  .=") for _ in range(rng.randint(1, 7))) for _ in range(rng.randint(0, 12))]
        line = " ".join(toks)
        if rng.random() < 0.01:
            line = line[: len(line) // 2] + "\0" + line[len(line) // 2:]
        out += (line + "\n").encode()
    for at in (1021, 1022, 1023, 16381, 16382, 16383, 32766, 65534):
        out[at:at + 3] = words[at % len(words)].encode()
    data = bytes(out[:390001]) + words[7].encode()
    for bs in (262140, 3000):
        want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=bs)
        assert stats.n_lines == nlines and got == want, bs
    assert len(want) > 20000


@pytest.mark.gpu
def test_expressions_that_can_never_match(torch_cuda):
    """Contradictory assertions (an automaton without nodes) next to ordinary always-on and literal-anchored expressions:
    found by tools/fuzz_gpu.py — the kernels used to stage the neighbouring pattern's tables for the empty one."""
    rng = random.Random(41)
    pats = ["\\b\\Bc", "[^a]-$", "\\B\\b$(1){1}", "needle_long", "x\\b\\By", "[^a]-", "zq"]
    flags = [7, 10, 15, 14, 6, 10, 10]
    ids = [2, 3, 2, 0, 1, 4, 5]
    data = regex_gen.random_text(rng, 3000, final_newline=True) + b"needle_long c- 1\nzq x y\n" * 20
    for bs in (262140, 64):
        want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=bs)
        assert stats.n_lines == nlines and got == want, bs
    assert {h[1] for h in want} == {0, 3, 4, 5}


@pytest.mark.gpu
def test_maximum_buffer_size(torch_cuda, tmp_path):
    """buffer_size = INT_MAX (the largest value the reference's int argument takes): pieces are whole lines, the end offset
    field of the sort key is 31 bits wide; through the device API and through the file API (a small file stays one chunk)."""
    import hypergrep_amd

    rng = random.Random(52)
    pats = ["foo", "ba+r", "x=[0-9]+;", "needle_in_haystack", "\\bend$"]
    flags = [14, 6, 14, 14, 14]
    ids = [0, 1, 2, 3, 0]
    lines = [" ".join(rng.choice(["foo", "bar", "baaar", "x=12;", "needle_in_haystack", "end", "lorem", "ipsum"]) for _ in range(rng.randint(0, 9))) for _ in range(4000)]
    lines[1234] = "z" * 70000 + " foo end"
    data = ("\n".join(lines) + "\n").encode()
    big = 2147483647
    want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=big)
    got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=big)
    assert stats.n_lines == nlines == 4000 and got == want and len(want) > 3000
    path = tmp_path / "max.log"
    path.write_bytes(data)
    want_rc, want_rows, want_batches = oracle_py.scan_file(str(path), pats, flags, ids, buffer_size=big, buffer_count=2)
    rows, batches = [], []

    def cb(matches, count):
        batches.append(count)
        rows.extend((matches[i].line_number, matches[i].id, matches[i].line) for i in range(count))

    rc = hypergrep_amd.scan(str(path), pats, cb, flags=flags, ids=ids, buffer_size=big, buffer_count=2)
    assert (rc, rows, batches) == (want_rc, want_rows, want_batches) and rc == 0


@pytest.mark.gpu
def test_concurrent_scans_from_threads(torch_cuda, tmp_path, monkeypatch):
    """The reference's parallel_grep runs one hyperscan() per file on a thread pool (multiscanner.py:197-209): many
    concurrent calls in one process, two pattern sets interleaved, files from empty to several ingest chunks — every
    call must deliver exactly what it delivers alone."""
    import concurrent.futures

    import hypergrep_amd

    monkeypatch.setenv("HYPERGREP_CHUNK_MB", "1")
    rng = random.Random(61)
    sets = [(["ERROR", "status=5[0-9]{2}", "needle_in_haystack"], [14, 14, 14], [0, 1, 2]),
            (["\\bGET\\b /api", "x=\\d+;", "(?i)warn"], [14, 6, 14], [5, 5, 6])]
    words = [b"ERROR", b"status=503", b"status=200", b"needle_in_haystack", b"GET /api", b"GETS /api", b"x=12;", b"Warn", b"lorem", b"ipsum", b"dolor"]
    files = []
    for i in range(24):
        nlines = rng.choice([0, 1, 50, 2000, 30000])
        body = b"".join(b" ".join(rng.choice(words) for _ in range(rng.randint(0, 8))) + b"\n" for _ in range(nlines))
        path = tmp_path / f"f{i}.log"
        path.write_bytes(body)
        files.append(str(path))

    def scan_one(job):
        path, (pats, flags, ids) = job
        rows, batches = [], []

        def cb(matches, count):
            batches.append(count)
            rows.extend((matches[i].line_number, matches[i].id, matches[i].line) for i in range(count))

        rc = hypergrep_amd.scan(path, pats, cb, flags=flags, ids=ids, buffer_count=rng.choice([1, 16, 64]))
        return rc, rows, sum(batches)

    jobs = [(f, sets[i % 2]) for i, f in enumerate(files)] + [(f, sets[(i + 1) % 2]) for i, f in enumerate(files)]
    alone = [scan_one(j) for j in jobs]
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        together = list(pool.map(scan_one, jobs))
    assert together == alone
    assert all(rc == 0 for rc, _, _ in alone) and sum(n for _, _, n in alone) > 50000
    # a few of them against the oracle
    for j in (3, 10, 30):
        path, (pats, flags, ids) = jobs[j]
        want_rc, want_rows, _ = oracle_py.scan_file(path, pats, flags, ids)
        assert (alone[j][0], alone[j][1]) == (want_rc, want_rows)


@pytest.mark.gpu
def test_skewed_hits_one_long_line_all_matches(torch_cuda):
    """All-matches expressions (no SINGLEMATCH) on one very long line: every occurrence of the required literal re-reports
    every match end of the line, so ONE confirm block stages millions of raw hits while the others stage none.  Equal
    per-block segments sized for the fullest block would need > 2^31 records (found by tools/fuzz_gpu.py: the call failed);
    full blocks now append to the compact array directly."""
    rng = random.Random(71)
    filler = "".join(rng.choice("abcx01 ._-") for _ in range(60000))
    long_line = "".join(filler[i:i + 1200] + " needle" for i in range(0, 60000, 1200))
    lines = [" ".join(rng.choice(["needle", "foo", "x=1", "lorem"]) for _ in range(rng.randint(0, 8))) for _ in range(2000)]
    lines[777] = long_line
    data = ("\n".join(lines) + "\n").encode()
    pats = ["needle.{2,}", "aa[a-z]", "foo"]
    flags = [6, 6, 14]
    ids = [0, 1, 2]
    want, nlines = oracle_hits(data, pats, flags, ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids)
    assert stats.n_lines == nlines and got == want
    assert len(want) > 50000 and stats.n_raw_hits > 1000000


@pytest.mark.gpu
def test_finalize_bucket_size_classes(torch_cuda):
    """Hits clustered on a few lines: buckets of the finalize with a few hundred reports (sorted by one block in LDS), with
    one to four thousand (sorted in a scratch area in HBM) and the ordinary ones (one wave, in registers) side by side;
    several expressions share ids, SINGLEMATCH and all-matches mixed, duplicates by construction."""
    rng = random.Random(73)
    lines = [" ".join(rng.choice(["needle7", "foo", "x=1", "lorem", "needle8x"]) for _ in range(rng.randint(0, 8))) for _ in range(3000)]
    lines[500] = " ".join("needle%d" % (i % 10) for i in range(300))
    lines[1500] = " ".join("needle%d%s" % (i % 10, "x" * (i % 3)) for i in range(900))
    lines[2500] = " ".join("needle7" for _ in range(1200))
    data = ("\n".join(lines) + "\n").encode()
    pats = ["needle[0-9]", "needle[0-9]x*", "foo", "needle7"]
    flags = [6, 6, 14, 14]
    ids = [0, 0, 2, 3]
    want, nlines = oracle_hits(data, pats, flags, ids)
    got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids)
    assert stats.n_lines == nlines and got == want
    assert len(want) > 8000 and stats.n_raw_hits > len(want)


@pytest.mark.gpu
def test_five_byte_literals_probe_every_second_byte(torch_cuda):
    """Every required literal of the set has >= 5 bytes: byte-aligned probing at even offsets only (two windows per literal,
    eight probes per 16 bytes).  Occurrences at every alignment, across rows, tiles and the end of the text, NULs, small buffers."""
    from hypergrep_amd import device

    pats = ["ERROR", "panic", "(?i)Failed", "denied: [a-z]+", "needle_in_haystack", "x=\\d+;", "\\bGET\\b /api"]
    flags = [14, 14, 14, 6, 14, 14, 14]
    ids = [0, 1, 2, 3, 4, 5, 5]
    info = device.Database(pats, flags, ids).info()
    assert info["byte_windows"] == 2 and info["n_always_on"] == 1 and info["n_windows"] == 2 * info["n_factors"]
    rng = random.Random(92)
    words = [b"ERROR", b"panic", b"FAILED", b"failed", b"denied: abc", b"needle_in_haystack", b"x=12;", b"ERRO", b"pani", b"faile", b"denied:",
             b"ERRORERROR", b"GET /api", b"GETS /api"]
    for trial in range(3):
        out = bytearray()
        while len(out) < 300000:
            line = bytearray()
            for _ in range(rng.choice([0, 2, 6, 6, 40])):
                line += rng.choice(words) if rng.random() < 0.4 else bytes(rng.choice(b"abcdefoOrRE =.;0123") for _ in range(rng.randint(1, 9)))
                if rng.random() < 0.5:
                    line += b" "
            if rng.random() < 0.02:
                line[len(line) // 2:len(line) // 2] = b"\0"
            out += line + b"\n"
        out[100000:100000] = b"z" * 30000 + b" ERROR panic " + b"y" * 20000
        for at in (1019, 1020, 1021, 1022, 1023, 2045, 16379, 16380, 16381, 16382, 16383, 32765, 65533):
            out[at:at + 5] = b"ERROR"
            out[at + 3000:at + 3005] = b"panic"
        data = bytes(out[:290000 + trial]) + rng.choice([b"panic", b"ERROR", b"ERROR\n", b"pani"])
        for bs in (262140, 4096):
            want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
            got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=bs)
            assert stats.n_lines == nlines and got == want, (trial, bs)
        assert len(want) > 3000


# ------------------------------------------------------------------ Face A artefact under the shim's call sequence
def _call_order_driver() -> str:
    import subprocess

    src = os.path.join(HERE, "native", "hs_call_order.c")
    exe = os.path.join(HERE, "native", "hs_call_order")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-o", exe, src, "-ldl"])
    return exe


def test_face_a_artefact_under_the_shim_call_order(torch_cuda, tmp_path):
    """hypergrep_amd/lib/libhs.so.5 (SONAME libhs.so.5) driven by tests/native/hs_call_order.c, which replays the reference
    shim's use of libhs (hyperscanner.c:136,140,301,217 per line piece,323,324 incl. the NULL frees), against the oracle's
    libhs.so.5 driven by the same program: identical reports on the reference's data files and on a seeded text."""
    import subprocess

    exe = _call_order_driver()
    product = os.path.join(os.path.dirname(HERE), "hypergrep_amd", "lib", "libhs.so.5")
    oracle = os.path.join(oracle_py.ORACLE_DIR, "_build", "libhs.so.5")
    assert os.path.exists(product) and os.path.exists(oracle)
    rng = random.Random(77)
    pairs = [regex_gen.anchored_pattern(rng) for _ in range(6)]
    seeded = tmp_path / "seeded.txt"
    seeded.write_bytes(regex_gen.anchored_text(rng, [s for _, s in pairs], 300) + b"\0\0lead " + b"x" * 300 + b"\nfoo\0bar\n")
    cases = [
        (os.path.join(FILES, "samplefile.txt"), 262140, [(14, 0, "bar"), (14, 1, "foo")]),
        (os.path.join(FILES, "greptest1.txt"), 262140, [(14, 0, "[a-z]+ing\\b"), (6, 1, "the"), (15, 2, "THE")]),
        (os.path.join(FILES, "greptest2.txt"), 64, [(14, 0, "\\d+"), (10, 3, "^.{3,8}$")]),
        (str(seeded), 128, [(14, i, p) for i, (p, _) in enumerate(pairs)] + [(6, 9, "foo"), (14, 9, "x{5}")]),
    ]
    total = 0
    for path, bs, pats in cases:
        if not os.path.exists(path):
            pytest.fail(f"fixture missing: {path}")
        args = [str(x) for t in pats for x in t]
        want = subprocess.run([exe, oracle, path, str(bs), "--"] + args, capture_output=True, text=True, timeout=120)
        got = subprocess.run([exe, product, path, str(bs), "--"] + args, capture_output=True, text=True, timeout=300)
        assert want.returncode == 0, want.stderr
        assert got.returncode == 0, got.stderr
        # inside one hs_scan call the oracle reports by ascending end offset, ties by id — so does the product
        assert got.stdout == want.stdout, (path, pats)
        total += len(want.stdout.splitlines())
    assert total > 100
    # a rejected expression: HYPERSCANNER_COMPILE (2) through the same sequence, error record freed
    bad = subprocess.run([exe, product, cases[0][0], "64", "--", "14", "0", "(?<=a)b"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "compile failed" in bad.stderr


# ------------------------------------------------------------------ regressions for round 1's GPU memory fault
def _run_gpu_cases(mode: str, env: dict | None = None, timeout: int = 600) -> dict:
    import subprocess
    import sys

    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(HERE, "gpu_cases.py"), mode], capture_output=True, text=True, timeout=timeout, env=e)
    assert out.returncode == 0, f"gpu_cases.py {mode} failed (rc {out.returncode}):\n{out.stdout[-2000:]}\n{out.stderr[-3000:]}"
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_no_kernel_reads_past_the_text(torch_cuda):
    """Parity cases on GUARDED text buffers (the address space right after the text is unmapped): zero bytes past a partial
    last tile that look like spaces or like a zero window, byte-aligned and dword-aligned windows, filters of every size
    class, always-on patterns, small scan buffers.  Round 1's fault was the drain of the stream kernel reading neighbour
    bytes of window positions PAST the text (tests/gpu_cases.py::guarded_cases has the case); in a child process, because a
    violation is a GPU memory fault."""
    res = _run_gpu_cases("guarded")
    assert res["ok"] and res["cases"] >= 70, res


@pytest.mark.parametrize("name", ["c2", "c3", "c5"])
def test_benchmark_workloads_against_python_re(torch_cuda, name):
    """A second checker that is NOT the oracle: the HIP path on 64 MiB of each benchmark workload against Python `re` run per
    expression over the same bytes (tests/re_check.py), compared as sets of (line, expression); line count against a byte
    count.  The CPU side of the same check (oracle vs `re`): tests/test_re_crosscheck.py."""
    import re_check
    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    spec = {"c2": benchspec.c2_spec, "c3": benchspec.c3_spec, "c5": benchspec.c5_spec}[name]
    patterns, needles, hpm = spec()
    ids = list(range(len(patterns)))
    nbytes = 64 << 20
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + int(name[1]), needles, hpm)
    torch.cuda.synchronize()
    host = bytes(text[:nbytes].cpu().numpy())
    sc = device.Scanner(device.Database(patterns, ids=ids), 0)
    st = sc.scan(text.data_ptr(), nbytes)
    got = {(h[0], h[1]) for h in sc.hits()}
    want = re_check.literal_line_id_pairs(host, patterns) if name == "c5" else re_check.line_id_pairs(host, patterns)
    assert st.n_lines == host.count(b"\n") + (0 if host.endswith(b"\n") else 1)
    assert got == want, (len(got), len(want), sorted(got - want)[:5], sorted(want - got)[:5])
    assert len(want) > {"c2": 3000, "c3": 3000, "c5": 40000}[name]
    assert _loaded_native()


def test_huge_patterns_match_oracle(torch_cuda):
    """Expressions of more than 1024 automaton positions ([a-z]{2000}x, .{0,3000}foo, foo.{0,3000}bar,
    (abc|def){200}, a{32767} gave rc 4): accept / reject as the oracle, and the wave-cooperative routine (hg_huge.hip)
    against the oracle on texts with hits, near misses and lines longer than the repeat — buffer API on guarded buffers,
    file API, block mode.  In a child process: new kernels, and a violation would be a GPU memory fault."""
    res = _run_gpu_cases("huge", timeout=900)
    assert res["ok"] and res["cases"] >= 30 and res["block_events"] >= 5, res


def test_many_pattern_sets_through_the_file_api(torch_cuda):
    """48 distinct pattern sets through hyperscan() in ONE process, once with room for every context to stay alive (the
    state round 1's fault needed) and once with a pool of 4, which makes contexts change their pattern set (scanner
    replaced, staging buffers kept)."""
    res = _run_gpu_cases("many-sets", {"HYPERGREP_POOL": "64"})
    assert res["ok"] and res["sets"] == 48 and res["contexts_alive"] >= 40, res
    res = _run_gpu_cases("many-sets", {"HYPERGREP_POOL": "4"})
    assert res["ok"] and res["contexts_alive"] <= 4 and res["contexts_rebound"] >= 40, res


# ------------------------------------------------------------------ BASELINE configs 1, 2 and 5 (config 3: above; config 4: 8 GPUs)
def test_config1_one_literal_one_mib_through_grep(torch_cuda, tmp_path):
    """BASELINE config 1: one literal pattern over a 1 MiB plaintext file through grep() (hypergrep/utils.py:147-231), the
    whole drop-in path: file -> hyperscan() -> batches -> (1-based line number, decoded line) tuples."""
    import hypergrep_amd
    from hypergrep_amd import benchspec, device

    patterns, needles, hpm = benchspec.c1_spec()
    data = device.synth_host(1 << 20, benchspec.SEED_BASE + 1, needles, hpm)
    path = tmp_path / "c1.log"
    path.write_bytes(data)
    rows, rc = hypergrep_amd.grep(str(path), patterns)
    want_rc, want_rows, _ = oracle_py.scan_file(str(path), patterns)
    assert rc == want_rc == 0
    assert rows == [(ln + 1, line.decode()) for ln, _id, line in want_rows]
    assert 50 < len(rows) < 200  # 1 % of ~9 500 lines
    assert hypergrep_amd.grep(str(path), patterns, count_only=True) == (len(rows), 0)
    assert _loaded_native()


def test_config2_one_regex_four_gib(torch_cuda):
    """BASELINE config 2: one regex (character class + bounded repeat) over 4 GiB of synthetic log on one GPU: exact line
    count, strict order, idempotence, and the head of the text against the oracle."""
    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    patterns, needles, hpm = benchspec.c2_spec()
    nbytes = 4 << 30
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + 2, needles, hpm)
    torch.cuda.synchronize()
    newlines = int((text[:nbytes] == 10).sum())
    db = device.Database(patterns)
    sc = device.Scanner(db, 0)

    def run():
        st = sc.scan(text.data_ptr(), nbytes)
        buf = torch.empty((st.n_hits, 2), dtype=torch.int64, device="cuda:0")
        assert sc.copy_hits_to(buf.data_ptr(), st.n_hits) == st.n_hits
        torch.cuda.synchronize()
        return st, buf

    st, a = run()
    assert st.n_lines == newlines + (0 if int(text[nbytes - 1]) == 10 else 1)
    assert bool((a[1:, 0] > a[:-1, 0]).all())  # one id, SINGLEMATCH: one report per line, lines strictly increasing
    assert 0.008 < st.n_hits / st.n_lines < 0.012
    st2, b = run()
    assert st2.n_hits == st.n_hits and bool((a == b).all())
    head_n = 2 << 20
    host = bytes(text[:head_n].cpu().numpy())
    head_n = host.rfind(b"\n") + 1
    want, nl = oracle_hits(host[:head_n], patterns)
    got = [h[:3] for h in sc.hits(limit=len(want) + 16) if h[0] < nl]
    assert got == [w[:3] for w in want]


def test_config5_4096_literals_use_the_wide_filter(torch_cuda):
    """BASELINE config 5 (single-GPU side): 4096 literals, 10 % of the lines hit.  The set needs the wide (two-cell, cuckoo)
    LDS filter; 64 MiB against properties + the head against the oracle."""
    import ctypes

    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    patterns, needles, hpm = benchspec.c5_spec()
    ids = list(range(len(patterns)))
    db = device.Database(patterns, ids=ids)
    info = db.info()
    assert info["n_patterns"] == 4096 and info["n_always_on"] == 0 and info["n_windows"] >= 4 * 4096
    assert info["byte_windows"] == 0  # dword-aligned windows; > 14 000 of them: the wide filter (hg_compile.cpp build_filter)
    nbytes = 64 << 20
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + 5, needles, hpm)
    torch.cuda.synchronize()
    sc = device.Scanner(db, 0)
    st = sc.scan(text.data_ptr(), nbytes)
    newlines = int((text[:nbytes] == 10).sum())
    assert st.n_lines == newlines + (0 if int(text[nbytes - 1]) == 10 else 1)
    assert 0.08 < st.n_hits / st.n_lines < 0.12
    head_n = 1 << 20
    host = bytes(text[:head_n].cpu().numpy())
    head_n = host.rfind(b"\n") + 1
    want, nl = oracle_hits(host[:head_n], patterns, ids=ids)
    got = [h[:3] for h in sc.hits(limit=len(want) + 64) if h[0] < nl]
    assert got == [w[:3] for w in want]
    del ctypes


def test_config5_full_size_properties_32gib(torch_cuda):
    """BASELINE config 5 at its per-GPU size (4096 literals, 10 % of the lines hit, 32 GiB in HBM): size-independent
    properties — exact line count, strict (line, id, to) order, every report's `to` consistent with a 13-byte literal,
    hit rate, idempotence — the head of the text against the oracle, and a window further in against Python `re`."""
    import re_check
    from hypergrep_amd import benchspec, device

    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < 90 << 30:
        pytest.skip("needs 90 GiB of free HBM")
    patterns, needles, hpm = benchspec.c5_spec()
    ids = list(range(len(patterns)))
    nbytes = 32 << 30
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + 5, needles, hpm)
    torch.cuda.synchronize()
    newlines = sum(int((text[o:min(o + (4 << 30), nbytes)] == 10).sum()) for o in range(0, nbytes, 4 << 30))
    db = device.Database(patterns, ids=ids)
    db.tune(b"".join(bytes(text[(i * (64 << 20)) & ~15: ((i * (64 << 20)) & ~15) + (256 << 10)].cpu().numpy()) for i in range(4)))
    sc = device.Scanner(db, 0)

    def run():
        st = sc.scan(text.data_ptr(), nbytes)
        buf = torch.empty((st.n_hits, 2), dtype=torch.int64, device="cuda:0")
        assert sc.copy_hits_to(buf.data_ptr(), st.n_hits) == st.n_hits
        torch.cuda.synchronize()
        return st, buf

    st, a = run()
    assert st.n_lines == newlines + (0 if int(text[nbytes - 1]) == 10 else 1)
    assert st.stream_launches == 4
    line, rest = a[:, 0], a[:, 1]
    ident, to = rest & 0xFFFFFFFF, (rest >> 32) & 0xFFFFFFFF
    key = (ident << 32) | to
    same_line = line[1:] == line[:-1]
    assert bool(((line[1:] > line[:-1]) | (same_line & (key[1:] > key[:-1]))).all())
    assert int(ident.max()) < 4096 and int(to.min()) >= 13  # a match ends at least a literal's length into its line
    assert 0.09 < st.n_hits / st.n_lines < 0.11  # 10 % of the lines carry a needle
    st2, b = run()
    assert st2.n_hits == st.n_hits and bool((a == b).all())  # idempotent
    del b
    # the head against the oracle
    head_n = 1 << 20
    host = bytes(text[:head_n].cpu().numpy())
    head_n = host.rfind(b"\n") + 1
    want, nl = oracle_hits(host[:head_n], patterns, ids=ids)
    n_head = int((line < nl).sum())
    got = [(int(r[0]), int(r[1]) & 0xFFFFFFFF, int(r[1]) >> 32) for r in a[:n_head].cpu().tolist()]
    assert got == [w[:3] for w in want]
    # 8 MiB in the third pipeline chunk against Python re: the lines that begin in the window, by global line number
    at = (20 << 30) + 12345
    win = bytes(text[at: at + (8 << 20)].cpu().numpy())
    first_nl = win.find(b"\n") + 1
    win = win[first_nl: win.rfind(b"\n") + 1]
    lines_before = sum(int((text[o:min(o + (4 << 30), at + first_nl)] == 10).sum()) for o in range(0, at + first_nl, 4 << 30))
    want_pairs = {(ln + lines_before, i) for ln, i in re_check.literal_line_id_pairs(win, patterns)}
    lo, hi = lines_before, lines_before + win.count(b"\n")
    sel = a[(line >= lo) & (line < hi)].cpu().tolist()
    assert {(int(r[0]), int(r[1]) & 0xFFFFFFFF) for r in sel} == want_pairs and len(want_pairs) > 5000


def test_parallel_grep_with_worker_processes(torch_cuda, tmp_path, capsys):
    """The --mp path of the reference's CLI (hypergrep/multiscanner.py:197-198, 538-543: a multiprocessing.Pool instead of
    threads): here spawned workers, each initialising the GPU runtime itself.  Output equals the threaded run's."""
    from hypergrep_amd import multiscanner

    rng = random.Random(5)
    files = []
    for i in range(3):
        p = tmp_path / f"f{i}.log"
        p.write_bytes(regex_gen.random_text(rng, 400, maxlen=60) + b"needle_in_haystack %d\n" % i)
        files.append(str(p))
    rc_threads = multiscanner.parallel_grep(files, ["needle_in_haystack", "a[bc]+x"], ordered_results=True, with_file_name=True, with_line_number=True)
    out_threads = capsys.readouterr().out
    rc_procs = multiscanner.parallel_grep(files, ["needle_in_haystack", "a[bc]+x"], ordered_results=True, with_file_name=True, with_line_number=True,
                                          use_multithreading=False)
    out_procs = capsys.readouterr().out
    assert rc_threads == rc_procs == 0
    assert out_procs == out_threads and out_threads.count("needle_in_haystack") == 3


def test_always_on_expressions_share_a_state_word(torch_cuda):
    """Always-on expressions (no required literal of 3 bytes): the context-free single-word ones are packed into shared state
    words (HgSlowGroup: one pass advances all of them) — mixed with SINGLEMATCH and all-matches flags, shared ids, an
    expression with boundary conditions (runs on its own), an unbounded one (the group's lead-in becomes the line start),
    a prefiltered one; small scan buffers force piece breaks through the packed automata."""
    rng = random.Random(808)
    pats = ["[a-z]+@[a-z]+", "x[0-9]+y", "[A-Z]{3}-[0-9]{4}:", "\\bq[a-z]*z\\b", "[0-9]+\\.[0-9]+", "=7", "needle_in_haystack_[0-9]+", "[_-][xyz]?[019]", "a{2,3}b"]
    flags = [14, 6, 14, 14, 6, 14, 14, 6, 14]
    ids = [0, 1, 1, 2, 3, 3, 4, 0, 5]
    words = ["joe@host", "x12y", "ABC-1234:", "qz", "quiz", "3.14", "=7", "needle_in_haystack_42", "_x0", "-9", "aab", "aaab", "plain", "x1", "AB-12:", "q z"]
    lines = []
    for _ in range(3000):
        n = rng.randint(0, 7)
        lines.append(" ".join(rng.choice(words) if rng.random() < 0.4 else "".join(rng.choice("abqxyzABC019.@=-_: ") for _ in range(rng.randint(1, 9))) for _ in range(n)))
    data = ("\n".join(lines) + "\n").encode()
    data = data[:40000] + b"\0" + data[40000:70000] + b"x" * 3000 + b"12y joe@" + b"h" * 2500 + b"\n" + data[70000:]
    for bs in (262140, 1000, 64):
        want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=bs)
        assert stats.n_lines == nlines
        assert got == want, bs
        assert len(want) > 1000
    # one group only (its tables stay staged), and a group next to a generic (multi-word) always-on expression
    for sub in ([0, 1, 2], [0, 1, 4, 5]):
        p, f, i = [pats[k] for k in sub], [flags[k] for k in sub], [ids[k] for k in sub]
        want, nlines = oracle_hits(data, p, f, i)
        got, stats = gpu_scan_buffer(torch_cuda, data, p, f, i)
        assert got == want and stats.n_lines == nlines


def test_always_on_groups_with_boundary_conditions(torch_cuda, monkeypatch):
    """Always-on expressions with boundary conditions (\\b \\B ^ $, with and without MULTILINE) share state words too: their
    groups carry the union of the per-context tables, and context-free expressions ride along.  More expressions than one
    word holds (several groups), SINGLEMATCH and all-matches members, shared ids, NUL bytes, over-long lines, small scan
    buffers; the same set without the packing must give the same hits."""
    from hypergrep_amd import device

    rng = random.Random(909)
    pats = ["\\bq[a-z]*z\\b", "^[A-Z]", "[0-9]$", "\\b[0-9]{3}\\b", "\\Bz", "x[0-9]+y", "=7", "^ab", "\\b[a-c]{2}\\b", "[xyz]$", "\\bq\\B", "^[0-9]+\\.", "[_-][xyz]?[019]", "\\b[A-Z]{2}-",
            "[a-z]+@[a-z]+"]
    flags = [14, 14, 6, 14, 6, 14, 14, 10, 6, 14, 14, 6, 6, 14, 14]
    ids = [0, 1, 2, 2, 3, 4, 5, 6, 7, 7, 8, 9, 0, 10, 11]
    words = ["joe@host", "x12y", "AB-12:", "qz", "quiz", "3.14", "=7", "_x0", "-9", "ab", "cab", "123", "1234", "q", "qq", "Zz", "z", "xyz", "7"]
    lines = []
    for _ in range(3000):
        n = rng.randint(0, 7)
        lines.append(" ".join(rng.choice(words) if rng.random() < 0.5 else "".join(rng.choice("abqxyzABC019.@=-_: ") for _ in range(rng.randint(1, 9))) for _ in range(n)))
    data = ("\n".join(lines) + "\n").encode()
    data = data[:30000] + b"\0" + data[30000:60000] + b"q" * 3000 + b"z 123 " + b"h" * 2500 + b"9\n" + data[60000:] + b"ab 12"
    db = device.Database(pats, flags, ids)
    info = db.info()
    assert info["n_always_on"] == len(pats) and info["max_state_words"] == 1
    for bs in (262140, 1000, 64):
        want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
        got, stats = gpu_scan_buffer(torch_cuda, data, pats, flags, ids, buffer_size=bs)
        assert stats.n_lines == nlines
        assert got == want, bs
        assert len(want) > 3000
    want, _ = oracle_hits(data, pats, flags, ids)
    monkeypatch.setenv("HG_NO_CTX_GROUPS", "1")
    got, _ = gpu_scan_buffer(torch_cuda, data, pats, flags, ids)
    assert got == want
    monkeypatch.delenv("HG_NO_CTX_GROUPS")
    # a single group with conditions (tables stay staged), and one expression with conditions beside a context-free group
    for sub in ([0, 3, 5], [3, 5, 6, 14], [1, 2]):
        p, f, i = [pats[k] for k in sub], [flags[k] for k in sub], [ids[k] for k in sub]
        want, nlines = oracle_hits(data, p, f, i)
        got, stats = gpu_scan_buffer(torch_cuda, data, p, f, i)
        assert got == want and stats.n_lines == nlines, sub


def test_scan_in_segments(torch_cuda, monkeypatch):
    """More reports than one pass may hold (2^28; lowered here): the buffer is scanned in 2, 4, 8 ... segments, each pass
    reporting the pieces that start in its stretch and scanning on as far as such a piece reaches; the segments' ordered
    hits are put one after the other.  Lines longer than the scan buffer (pieces) across segment boundaries, NUL bytes,
    always-on and prefiltered expressions, SINGLEMATCH and all-matches reports, a line base: equal to the oracle, in order."""
    from hypergrep_amd import device

    rng = random.Random(4242)
    pats = ["needle_in_haystack", "fo+bar[0-9]*", "a.c", "\\bGET\\b", "=7", "st.*us=2", "[0-9]+\\.[0-9]+", "(?i)error.*timeout"]
    flags = [14, 14, 6, 10, 14, 14, 6, 14]
    ids = [0, 1, 2, 3, 3, 4, 5, 1]
    parts = []
    for k in range(12):  # ~6 MiB: ordinary lines, and every 512 KiB or so a line of 300 000 bytes (two pieces at the default scan buffer)
        parts.append(regex_gen.random_text(rng, 4500, maxlen=200))
        parts.append(b"status=200 GET /a.c " + b"q" * 150000 + b" needle_in_haystack foobar7 abc =7 " + b"z" * 150000 + b" 3.14 ERROR x timeout\n")
    data = b"".join(parts)
    data = data[:700000] + b"\0" + data[700001:2000000] + b"\0\0" + data[2000002:]
    for bs, limit in ((262140, 6000), (3000, 2500), (262140, 40000)):
        want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
        monkeypatch.setenv("HG_HIT_LIMIT", str(limit))
        n = len(data)
        buf = torch_cuda.zeros(n + 32, dtype=torch_cuda.uint8, device="cuda:0")
        buf[:n] = torch_cuda.frombuffer(bytearray(data), dtype=torch_cuda.uint8).cuda()
        torch_cuda.cuda.synchronize()
        sc = device.Scanner(device.Database(pats, flags=flags, ids=ids), 0)
        stats = sc.scan(buf.data_ptr(), n, buffer_size=bs, line_base=5)
        got = sc.hits()
        assert stats.n_lines == nlines and stats.stream_launches >= 2, (bs, limit, stats)
        assert got == sorted(got)
        assert [(ln - 5, i, to, st, le) for ln, i, to, st, le in got] == want, (bs, limit)
        assert len(want) > limit
        monkeypatch.delenv("HG_HIT_LIMIT")
        # and the scanner is as good as new for an ordinary pass
        st2 = sc.scan(buf.data_ptr(), n, buffer_size=bs)
        assert sorted(sc.hits()) == want and st2.n_lines == nlines


def test_dense_candidates_shrink_the_pipeline_chunks(torch_cuda, monkeypatch):
    """A text whose every dword is a candidate: the workspace holds ONE pipeline chunk's candidates, and when that would pass
    the limit (2^30 records; lowered here) the engine halves the chunks instead of failing with "split the buffer".
    96 MiB of "aaaaaaaaa\\n" lines against the literal "aaaaaaaa": one hit per line, in order."""
    from hypergrep_amd import device

    torch = torch_cuda
    line = b"aaaaaaaaa\n"
    nbytes = 96 << 20
    reps = nbytes // len(line)
    nbytes = reps * len(line)
    text = torch.frombuffer(bytearray(line * 4096), dtype=torch.uint8).cuda().repeat(reps // 4096 + 1)[:nbytes].contiguous()
    pad = torch.zeros(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    pad[:nbytes] = text
    torch.cuda.synchronize()
    monkeypatch.setenv("HG_CAND_LIMIT", str(12 << 20))
    monkeypatch.setenv("HG_CHUNK_TILES", str(4096))  # the pipeline starts with 64 MiB chunks
    db = device.Database(["aaaaaaaa"])
    sc = device.Scanner(db, 0)
    st = sc.scan(pad.data_ptr(), nbytes)
    assert st.n_lines == reps and st.n_hits == reps and st.reruns >= 2
    assert st.stream_launches >= 3  # 96 MiB in chunks of 32 MiB or less
    buf = torch.empty((st.n_hits, 2), dtype=torch.int64, device="cuda:0")
    sc.copy_hits_to(buf.data_ptr(), st.n_hits)
    torch.cuda.synchronize()
    assert bool((buf[:, 0] == torch.arange(reps, device="cuda:0")).all())
    assert bool((buf[:, 1] >> 32 == 8).all())  # `to` = 8, id 0
    st2 = sc.scan(pad.data_ptr(), nbytes)
    assert st2.n_hits == reps and st2.reruns == 0  # the scanner remembers the chunk size that fits
