"""Throughput of patterns without a long required literal (always-on tier) next to a prefiltered one; GPU box only."""
import sys, time
sys.path.insert(0, '/root/repo')
import os
import torch
import hypergrep_amd
if os.environ.get("HG_LIB"):  # an experiment / earlier build of the native library (hypergrep_amd/build.py HG_BUILD_OUT)
    hypergrep_amd.configure_libraries(libhs=os.path.abspath(os.environ["HG_LIB"]))
from hypergrep_amd import benchspec, device
patterns, needles, hpm = benchspec.c3_spec()
SETS = (["ERROR"], ["foo|bar"], ["status=5[0-9]{2}"], ["ERROR", "WARN", "panic", "fail"], ["[0-9]+\\.[0-9]+"], ["fail.*time"], ["\\bGET\\b"], ["[a-z]+@[a-z]+"], ["\\b[0-9]{3}\\b"], ["[0-9]+\\.[0-9]+", "[a-z]+@[a-z]+", "\\b[0-9]{3}\\b", "=7"],
        # always-on expressions that (almost) never match: the cost of the automaton pass itself, one pattern and four
        ["[a-z]+@[a-z]+"], ["[a-z]+@[a-z]+", "x[0-9]+y", "[A-Z]{3}-[0-9]{4}:", "\\bq[a-z]*z\\b"],
        # huge automata (sparse tables, hg_huge.hip): anchored by a literal (12, 13), always-on (14), among config 3's patterns (15)
        ["status=500.{0,3000}timeout"], ["foo.{0,3000}bar"], ["[a-z]{2000}x"], patterns + ["status=5[0-9]{2}.{0,2000}retry_budget"],
        # 16: an always-on expression whose match can include the newline (an accepting node that consumes it)
        ["[a-z]+@[a-z]+\\s"],
        # 17, 18: always-on expressions of two state words (33..64 positions): a SHA-1 in hex, a UUID
        ["[0-9a-f]{40}"], ["[0-9a-f]{8}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{12}"])
for pats in (SETS if len(sys.argv) < 3 else [SETS[int(sys.argv[2])]]):
    nbytes = int(sys.argv[1]) << 30 if len(sys.argv) > 1 else 1 << 30
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, 77, needles, hpm)
    torch.cuda.synchronize()
    db = device.Database(pats, ids=list(range(len(pats))))
    print(pats, db.info())
    sc = device.Scanner(db, 0)
    sc.scan(text.data_ptr(), nbytes)
    t = time.perf_counter()
    for _ in range(3):
        st = sc.scan(text.data_ptr(), nbytes)
    dt = (time.perf_counter() - t) / 3
    print(f"  {nbytes / (1 << 30) / dt:8.1f} GiB/s  hits={st.n_hits} lines={st.n_lines} cands={st.n_candidates} stream={st.ms_stream:.3f} ms ({nbytes / 1e9 / max(st.ms_stream, 1e-9):.0f} GB/s) total={st.ms_total:.3f} ms")
