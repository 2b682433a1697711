#!/usr/bin/env python3
"""Regenerate tests/golden/*.json — BUILD CONTAINER ONLY (reads /root/reference; never runs on the GPU box).

What it does
------------
1. Imports the reference's own Python package from /root/reference and points its ctypes libhs loader
   (hypergrep/utils.py:125-144 configure_libraries) at oracle/_build/libhs.so.5 — the oracle's regex engine
   behind the six libhs symbols.  The reference's *unmodified prebuilt* shim
   (hypergrep/lib/libhyperscanner.so) and *unmodified* Python then run every engine-touching case of the
   reference's own test tables (hypergrep/test/test_hypergrep.py: check_hyperscan_compatibility, scan,
   grep, parallel_grep) and the result is asserted equal to the table's expectation.  That pins the
   oracle's regex semantics + the libhs boundary against the reference's golden expectations.
2. Captures plumbing vectors A-L through the same stack: byte-level inputs, the
   (line_number, line bytes) sequence and the batch sizes the reference shim produces.
3. Writes inputs + expected outputs as JSON fixtures (data only, no reference source text) and copies the
   reference's test data files into tests/golden/files/.

Intel Hyperscan itself is absent (.MISSING_LARGE_BLOBS), so constructs the reference's tables never
exercise stay "parity unpinned" against the real engine; see DESIGN.md.
"""
from __future__ import annotations

import base64
import contextlib
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
REF_TESTS = os.path.join(REF, "hypergrep", "test")
FILES = os.path.join(HERE, "files")


def b64(b: bytes) -> str:
    return base64.b64encode(b).decode()


def main() -> None:
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)
    libhs = os.path.join(REPO, "oracle", "_build", "libhs.so.5")

    sys.path.insert(0, REF)
    import hypergrep  # the reference package
    from hypergrep import multiscanner, utils

    hypergrep.configure_libraries(libhs=libhs)

    # Reference test data files -> fixtures (data, not source).
    os.makedirs(FILES, exist_ok=True)
    for name in ("samplefile.txt", "samplefile.txt.gz", "samplefile.txt.zst", "greptest1.txt", "greptest2.txt"):
        shutil.copyfile(os.path.join(REF_TESTS, name), os.path.join(FILES, name))

    # The reference's own case tables (pure data: args / kwargs / expected).
    sys.path.insert(0, REF_TESTS)
    sys.path.insert(0, REF)
    import test_hypergrep as ref_tests  # noqa: E402

    def rel(path: str) -> str:
        return path.replace(REF_TESTS + "/", "") if isinstance(path, str) else path

    table_out = {"check_compatibility": [], "scan": [], "grep": [], "parallel_grep": []}

    for name, case in ref_tests.TEST_CASES["check_hyperscan_compatibility"].items():
        got = utils.check_compatibility(*case["args"])
        assert got == case["returns"], (name, got)
        table_out["check_compatibility"].append({"name": name, "patterns": case["args"][0], "returns": case["returns"]})

    for name, case in ref_tests.TEST_CASES["scan"].items():
        path, patterns, _cb = case["args"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            utils.scan(path, patterns, ref_tests._basic_callback)
        got = buf.getvalue().splitlines()
        assert got == case["returns"], (name, got, case["returns"])
        table_out["scan"].append({"name": name, "file": rel(path), "patterns": patterns, "returns": case["returns"]})

    for name, case in ref_tests.TEST_CASES["grep"].items():
        path, patterns = case["args"]
        kwargs = case.get("kwargs", {})
        entry = {"name": name, "file": rel(path), "patterns": patterns, "kwargs": kwargs}
        if "raises" in case:
            try:
                utils.grep(path, patterns, **kwargs)
            except case["raises"]:
                pass
            else:
                raise AssertionError(name)
            entry["raises"] = case["raises"].__name__
        else:
            got = utils.grep(path, patterns, **kwargs)
            assert got == case["returns"], (name, got)
            entry["returns"] = [[list(t) for t in got[0]], got[1]]
        if entry["file"] == REF_TESTS:
            entry["file"] = "."
        table_out["grep"].append(entry)

    for name, case in ref_tests.TEST_CASES["parallel_grep"].items():
        files, patterns = case["args"]
        kwargs = case.get("kwargs", {})
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rc = multiscanner.parallel_grep(files, patterns, **kwargs)
        got = [line.replace(f"{REF_TESTS}/", "") for line in buf.getvalue().splitlines()]
        assert (got, rc) == case["returns"], (name, got, rc, case["returns"])
        table_out["parallel_grep"].append(
            {"name": name, "files": [rel(f) for f in files], "patterns": patterns, "kwargs": kwargs,
             "returns": [case["returns"][0], case["returns"][1]]}
        )

    with open(os.path.join(HERE, "reference_tables.json"), "w", encoding="utf-8") as out:
        json.dump(table_out, out, indent=1, sort_keys=True)

    # ---- the CLI layer (multiscanner.py): argument handling and pattern conversions, as input -> output vectors ----
    import shlex

    cli = {"argv": [], "to_basic_regular_expressions": [], "to_gnu_regular_expressions": []}
    argvs = ["pattern1 file1 file2 file3", "pattern1 -e pattern2 file1", "pattern1 -f regex.txt file1", "pattern1 -e pattern2 -f regex.txt file1",
             "-e pattern2 pattern1 -e pattern3 file1 file2", "pattern1 file1 -e pattern2 file2 -e pattern3 file3 f4", "p1 f1 f2 f3", "-e p2 p1 -e p3 f1 f2",
             "p1 f1 -e p2 f2 -e p3 f3 f4", "-i -n -H foo a b", "-c -h foo a", "-E -o 'a|b' x", "-P -q foo x y", "-l -s -m 3 foo x", "-L foo x", "-t --no-order --no-sort --mp foo b a",
             "--no-gnu -G foo x", "-a foo x", "foo"]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            with open("regex.txt", "wt", encoding="utf-8") as f:
                f.write("filepattern1\nfilepattern2")
            for argv in argvs:
                ns = multiscanner.parse_args(shlex.split(argv))
                attrs = {k: v for k, v in vars(ns).items() if k != "parser"}
                cli["argv"].append({"argv": argv, "attributes": attrs, "files": multiscanner.get_argparse_files(ns),
                                    "patterns": multiscanner.get_argparse_patterns(ns)})
        finally:
            os.chdir(cwd)
    bre_inputs = [["test"], ["^test.*[test]$"], ["^test.*[test]+?(){}|$"], [r"^test.*[test]+?(){}|\^\$\*\.\[\]\+\?\(\)\{\}\|$"],
                  [r"data \((?P<v0>.*?) (?P<v1>.*?)"], [r"a\|b", r"x\{2\}", "a{2}"], [r"foo\(bar\)\+", "(x)"], [r"tab\tq+", r"\w\+"], ["", "|"]]
    for inp in bre_inputs:
        try:
            cli["to_basic_regular_expressions"].append({"args": inp, "returns": multiscanner.to_basic_regular_expressions(inp)})
        except ValueError:
            cli["to_basic_regular_expressions"].append({"args": inp, "raises": "ValueError"})
    gnu_inputs = [["<foo>"], [r"<foo>\<foo\>"], [r"<foo>\<foo\>\\<foo\\>"], [r"\<a\>|\<b\>", "x"], [r"a\b\<"]]
    for inp in gnu_inputs:
        cli["to_gnu_regular_expressions"].append({"args": inp, "returns": multiscanner.to_gnu_regular_expressions(inp)})
    with open(os.path.join(HERE, "cli_tables.json"), "w", encoding="utf-8") as out:
        json.dump(cli, out, indent=1, sort_keys=True)
    print(f"cli tables: {len(cli['argv'])} argv cases, {len(bre_inputs)} BRE and {len(gnu_inputs)} GNU conversion cases from the reference's functions")
    n_cases = sum(len(v) for v in table_out.values())
    print(f"reference tables: {n_cases} engine-touching cases reproduced through the reference shim + oracle libhs")

    # ---- plumbing vectors A-L through the reference shim ----
    def run_scan(data: bytes | None, patterns, path=None, **kw):
        batches: list[int] = []
        rows: list[list] = []

        def cb(matches, count):
            batches.append(count)
            for i in range(count):
                m = matches[i]
                rows.append([m.line_number, m.id, b64(m.line)])

        if path is None:
            with tempfile.NamedTemporaryFile(delete=False) as tmp:
                tmp.write(data)
                path = tmp.name
            try:
                rc = utils.scan(path, patterns, cb, **kw)
            finally:
                os.unlink(path)
        else:
            rc = utils.scan(path, patterns, cb, **kw)
        return {"rc": rc, "rows": rows, "batches": batches}

    vectors = []

    def vec(name, data, patterns=("x",), **kw):
        res = run_scan(data, list(patterns), **kw)
        vectors.append({"name": name, "data": b64(data), "patterns": list(patterns), "kwargs": kw, **res})

    vec("A_last_line_without_newline", b"ax\nbx")
    vec("B_empty_lines", b"\n\nx\n\n")
    vec("C_crlf", b"x\r\nx\r\n")
    vec("D_long_line_split", b"0123456x89abcdefxh\nx\n", buffer_size=8)
    vec("E_exact_fit_split", b"x234567\nx\n", buffer_size=8)
    vec("F_interior_nul", b"ab\0x\nx\n")
    vec("G_leading_nul", b"\0\0x\nq\n")
    vec("H_max_match_2", b"x\nx\nx\nx\n", max_match_count=2)
    vec("I_max_3_buffer_2", b"x\nx\nx\nx\n", max_match_count=3, buffer_count=2)
    vec("J_empty_file", b"")
    vec("L_high_bytes", b"\xff\xfex\n")
    vec("M_batching_16_by_5", b"x\n" * 16, buffer_count=5)
    vec("N_two_ids_one_line", b"ab\nb\na\n", patterns=("a", "b"), ids=[7, 3])
    vec("O_shared_id_two_patterns", b"ab\nb\na\n", patterns=("a", "b"))
    vec("P_not_singlematch", b"aaa\nba\n", patterns=("a",), flags=[utils.HS_FLAG_DOTALL | utils.HS_FLAG_MULTILINE])
    vec("Q_max_count_overshoot_same_line", b"ab\nab\n", patterns=("a", "b"), ids=[1, 2], max_match_count=1)
    vec("R_nul_then_match_after_nul", b"\0a\0a\na\n", patterns=("a",))
    res = run_scan(None, ["x"], path="/nonexistent/definitely/missing")
    vectors.append({"name": "K_missing_file", "data": None, "patterns": ["x"], "kwargs": {}, **res})

    with open(os.path.join(HERE, "plumbing_vectors.json"), "w", encoding="utf-8") as out:
        json.dump(vectors, out, indent=1, sort_keys=True)
    for v in vectors:
        print(f"  {v['name']}: rc={v['rc']} rows={[(r[0], r[1], base64.b64decode(r[2])) for r in v['rows']]} batches={v['batches']}")


if __name__ == "__main__":
    main()
