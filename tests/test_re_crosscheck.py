"""The oracle against an independent checker on the ACTUAL benchmark workloads (hypergrep_amd/benchspec.py c2 / c3 / c5):
Python `re` per expression over the same synthetic text, compared as sets of (line, expression).  VERDICT r2 ran this by hand;
here it is committed.  The GPU side (the HIP path against the same `re`-derived sets, without the oracle in between) is
tests/test_gpu_parity.py::test_benchmark_workloads_against_python_re."""
from __future__ import annotations

import pytest

import oracle_py
import re_check
from hypergrep_amd import benchspec, device


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_oracle_equals_python_re_on_the_benchmark_specs(name):
    spec = {"c2": benchspec.c2_spec, "c3": benchspec.c3_spec}[name]
    patterns, needles, hpm = spec()
    text = device.synth_host(2 << 20, benchspec.SEED_BASE + int(name[1]), needles, hpm * 4)
    ids = list(range(len(patterns)))
    rc, hits, nlines = oracle_py.scan_buffer(text, patterns, ids=ids)
    assert rc == 0 and nlines == text.count(b"\n") + (0 if text.endswith(b"\n") else 1)
    got = {(h[0], h[1]) for h in hits}
    want = re_check.line_id_pairs(text, patterns)
    assert got == want
    assert len(want) > (50 if name == "c2" else 400)


def test_oracle_equals_substring_search_on_config5():
    patterns, needles, hpm = benchspec.c5_spec()
    text = device.synth_host(512 << 10, benchspec.SEED_BASE + 5, needles, hpm)
    ids = list(range(len(patterns)))
    rc, hits, _ = oracle_py.scan_buffer(text, patterns, ids=ids)
    assert rc == 0
    got = {(h[0], h[1]) for h in hits}
    want = re_check.literal_line_id_pairs(text, patterns)
    assert got == want and len(want) > 300
