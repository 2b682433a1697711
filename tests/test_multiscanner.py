"""CLI layer (hypergrep_amd/multiscanner.py) against vectors produced by the reference's own functions
(tests/golden/cli_tables.json, made by tests/golden/make_golden.py) and the reference's parallel_grep table
(tests/golden/reference_tables.json, test_hypergrep.py:292-909).  parallel_grep scans files, so it needs the GPU."""
import json
import os
import shlex

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = os.path.join(HERE, "golden", "files")
with open(os.path.join(HERE, "golden", "cli_tables.json"), encoding="utf-8") as _f:
    CLI = json.load(_f)
with open(os.path.join(HERE, "golden", "reference_tables.json"), encoding="utf-8") as _f:
    TABLES = json.load(_f)


@pytest.fixture()
def in_tmp_with_pattern_file(tmp_path, monkeypatch):
    (tmp_path / "regex.txt").write_text("filepattern1\nfilepattern2", encoding="utf-8")
    monkeypatch.chdir(tmp_path)


@pytest.mark.parametrize("case", CLI["argv"], ids=lambda c: c["argv"])
def test_parse_args_and_positional_rules(case, in_tmp_with_pattern_file):
    from hypergrep_amd import multiscanner

    ns = multiscanner.parse_args(shlex.split(case["argv"]))
    got = {k: v for k, v in vars(ns).items() if k != "parser"}
    assert got == case["attributes"]
    assert multiscanner.get_argparse_files(ns) == case["files"]
    assert multiscanner.get_argparse_patterns(ns) == case["patterns"]


@pytest.mark.parametrize("case", CLI["to_basic_regular_expressions"], ids=lambda c: repr(c["args"]))
def test_to_basic_regular_expressions(case):
    from hypergrep_amd import multiscanner

    if "raises" in case:
        with pytest.raises(ValueError, match="hyperscanner: invalid regex"):
            multiscanner.to_basic_regular_expressions(case["args"])
    else:
        assert multiscanner.to_basic_regular_expressions(case["args"]) == case["returns"]


@pytest.mark.parametrize("case", CLI["to_gnu_regular_expressions"], ids=lambda c: repr(c["args"]))
def test_to_gnu_regular_expressions(case):
    from hypergrep_amd import multiscanner

    assert multiscanner.to_gnu_regular_expressions(case["args"]) == case["returns"]


def test_pattern_errors_are_value_errors(in_tmp_with_pattern_file):
    from hypergrep_amd import multiscanner

    with pytest.raises(ValueError, match="invalid regex"):
        multiscanner.get_argparse_patterns(multiscanner.parse_args(["(unclosed", "f"]))
    with pytest.raises(ValueError, match="incompatible regex"):
        multiscanner.get_argparse_patterns(multiscanner.parse_args(["-P", "(?<!foo)bar", "f"]))


def test_print_results_and_read_stdin(capsys, monkeypatch):
    import io

    from hypergrep_amd import multiscanner

    rows = [(2, "foo\n"), (5, "bar\n")]
    multiscanner.print_results(rows, "f.txt")
    multiscanner.print_results(rows, "f.txt", with_file_name=True)
    multiscanner.print_results(rows, "f.txt", with_line_number=True)
    multiscanner.print_results(rows, "f.txt", with_file_name=True, with_line_number=True)
    multiscanner.print_results([], "f.txt", with_file_name=True)
    assert capsys.readouterr().out == "foo\nbar\nf.txt:foo\nf.txt:bar\n2:foo\n5:bar\nf.txt:2:foo\nf.txt:5:bar\n"
    monkeypatch.setattr("sys.stdin", io.StringIO("a.log\n  b.log  \n\nc.log\n"))
    assert list(multiscanner.read_stdin()) == ["a.log", "b.log"]


def test_main_usage_errors(capsys, monkeypatch):
    from hypergrep_amd import multiscanner

    for argv in (["hyperscanner"], ["hyperscanner", "(unclosed", "x"]):
        monkeypatch.setattr("sys.argv", argv)
        with pytest.raises(SystemExit) as exit_info:
            multiscanner.main()
        assert exit_info.value.code == 2
    out = capsys.readouterr().out
    # no pattern at all: the engine refuses an empty set (as Hyperscan does), reported like any other incompatible pattern
    assert "incompatible regex" in out and "invalid regex" in out


# ------------------------------------------------------------------ the reference's parallel_grep table (GPU: it scans files)
@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


@pytest.mark.gpu
@pytest.mark.parametrize("case", TABLES["parallel_grep"], ids=lambda c: c["name"])
def test_reference_table_parallel_grep(torch_cuda, case, capsys):
    from hypergrep_amd import multiscanner

    files = [os.path.join(FILES, name) if not os.path.isabs(name) else name for name in case["files"]]
    rc = multiscanner.parallel_grep(files, case["patterns"], **case["kwargs"])
    got = [line.replace(f"{FILES}/", "") for line in capsys.readouterr().out.splitlines()]
    assert [got, rc] == case["returns"]


@pytest.mark.gpu
def test_main_end_to_end(torch_cuda, capsys, monkeypatch):
    from hypergrep_amd import multiscanner

    f1, f2 = os.path.join(FILES, "greptest1.txt"), os.path.join(FILES, "greptest2.txt")
    monkeypatch.setattr("sys.argv", ["hyperscanner", "-c", "foo", f2, f1])
    with pytest.raises(SystemExit) as exit_info:
        multiscanner.main()
    assert exit_info.value.code == 0
    lines = capsys.readouterr().out.splitlines()
    assert [line.split(":")[0] for line in lines] == [f1, f2]  # sorted, file names shown for two files
    assert lines[0].endswith(":16")
    monkeypatch.setattr("sys.argv", ["hyperscanner", r"barfoo\+", f1])  # BRE: \+ is the operator
    with pytest.raises(SystemExit):
        multiscanner.main()
    assert capsys.readouterr().out == "barfoo\nbarfoo+\n"
