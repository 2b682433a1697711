#!/usr/bin/env python3
"""Per-wave timing of hg_confirm_fast_kernel (a library built with -DHG_PROFILE_CONFIRM writes one record per wave):
    HG_LIB=build/variants/prof.so python tools/confirm_waves.py [gib]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import hypergrep_amd
hypergrep_amd.configure_libraries(libhs=os.path.abspath(os.environ["HG_LIB"]))
import numpy as np, torch
from hypergrep_amd import benchspec, device
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8
patterns, needles, hpm = benchspec.c3_spec()
drop = [d for d in os.environ.get("CW_DROP", "").split(",") if d]
patterns = [p for p in patterns if not any(d in p for d in drop)]
print("patterns", len(patterns), "dropped", drop)
nbytes = int(gib * (1 << 30))
text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + 3, needles, hpm)
torch.cuda.synchronize()
db = device.Database(patterns, ids=list(range(len(patterns))))
db.tune(bytes(text[: 1 << 20].cpu().numpy()))
sc = device.Scanner(db, 0)
os.environ["HG_CHUNK_TILES"] = str(1 << 30)
for _ in range(3):
    st = sc.scan(text.data_ptr(), nbytes)
nw = 256 * 6 * 2 * 3 * 2
raw = ctypes.create_string_buffer(nw * 16)
assert device.lib().hg_debug_download(raw, ctypes.c_void_p(sc._last.d_hits + (6 << 20) * 16), nw * 16) == 0
rec = np.frombuffer(raw.raw, dtype=np.dtype([("t0", "<u8"), ("dur", "<u4"), ("tag", "<u4")]))
rec = rec[rec["t0"] > 0]
t0 = rec["t0"].min()
print("waves", len(rec), "kernel span us", (rec["t0"] + rec["dur"]).max() / 100.0 - t0 / 100.0, "hits", st.n_hits)
for m in range(3):
    r = rec[(rec["tag"] & 15) == m]
    if not len(r): continue
    d = r["dur"] / 100.0
    s = (r["t0"] - t0) / 100.0
    print(f"mode {m}: waves {len(r)}  dur us: mean {d.mean():.1f} p50 {np.percentile(d,50):.1f} p90 {np.percentile(d,90):.1f} p99 {np.percentile(d,99):.1f} max {d.max():.1f} | start us: p50 {np.percentile(s,50):.1f} p90 {np.percentile(s,90):.1f} max {s.max():.1f} | end max {(s+d).max():.1f}")

raw2 = ctypes.create_string_buffer(nw * 16)
assert device.lib().hg_debug_download(raw2, ctypes.c_void_p(sc._last.d_hits + ((6 << 20) + 32768) * 16), nw * 16) == 0
r2 = np.frombuffer(raw2.raw, dtype=np.dtype([("ls", "<u8"), ("run", "<u4"), ("tag", "<u4")]))
for m in (1, 2):
    r = r2[(r2["tag"] >> 28) == m]
    r = r[(r["tag"] & 0xFFFFFFF) > 0]
    if not len(r): continue
    load = (r["ls"] & 0xFFFFFFFF) / 100.0; stage = (r["ls"] >> 32) / 100.0; run = r["run"] / 100.0; pats = r["tag"] & 0xFFFFFFF
    print(f"mode {m}: busy waves {len(r)}: load us mean {load.mean():.1f} max {load.max():.1f}; stage mean {stage.mean():.1f} max {stage.max():.1f}; run mean {run.mean():.1f} p90 {np.percentile(run,90):.1f} max {run.max():.1f}; patterns per wave mean {pats.mean():.2f} max {pats.max()}")

nv = 12288 * 2
raw3 = ctypes.create_string_buffer(nv * 16)
assert device.lib().hg_debug_download(raw3, ctypes.c_void_p(sc._last.d_hits + ((6 << 20) + 65536) * 16), nv * 16) == 0
r3 = np.frombuffer(raw3.raw, dtype=np.dtype([("t0", "<u8"), ("dur", "<u4"), ("tag", "<u4")]))
r3 = r3[r3["t0"] > 0]
if len(r3):
    v0 = r3["t0"].min(); d = r3["dur"] / 100.0; st = (r3["t0"] - v0) / 100.0; rounds = r3["tag"] & 0xFF; pairs = r3["tag"] >> 8
    print(f"verify: waves {len(r3)} span us {(st + d).max():.1f}; dur mean {d.mean():.1f} p50 {np.percentile(d,50):.1f} p90 {np.percentile(d,90):.1f} max {d.max():.1f}; start p50 {np.percentile(st,50):.1f} p90 {np.percentile(st,90):.1f} max {st.max():.1f}; rounds per wave mean {rounds.mean():.2f} max {rounds.max()}; pairs per wave mean {pairs.mean():.1f} max {pairs.max()}")
