#!/usr/bin/env python3
"""End-to-end (Face B) rate: file on tmpfs -> hyperscan() -> callbacks.  PCIe- and IO-inclusive, never bench.py's `value`.
    python tools/e2e_bench.py [--gib 4] [--workload c3] [--reps 3]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=4.0)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dir", default="/dev/shm")
    args = ap.parse_args()
    import torch

    import hypergrep_amd
    from hypergrep_amd import benchspec, device

    patterns, needles, hpm = {"c1": benchspec.c1_spec, "c2": benchspec.c2_spec, "c3": benchspec.c3_spec, "c5": benchspec.c5_spec}[args.workload]()
    nbytes = int(args.gib * (1 << 30))
    path = os.path.join(args.dir, f"hg_e2e_{os.getpid()}.log")
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + 3, needles, hpm)
    torch.cuda.synchronize()
    with open(path, "wb") as f:
        step = 256 << 20
        for off in range(0, nbytes, step):
            f.write(text[off:min(off + step, nbytes)].cpu().numpy().tobytes())
    del text
    torch.cuda.empty_cache()
    try:
        for rep in range(args.reps):
            n = [0]

            def on_match(matches, count):
                n[0] += count

            t0 = time.perf_counter()
            rc = hypergrep_amd.scan(path, patterns, on_match, ids=list(range(len(patterns))), buffer_count=4096)
            dt = time.perf_counter() - t0
            print(f"rep {rep}: rc={rc} hits={n[0]} {dt:.3f} s  {nbytes / (1 << 30) / dt:.2f} GiB/s end to end", flush=True)
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
