#!/usr/bin/env python3
"""Call latency of the drop-in API (Face B) by file size: hypergrep_amd.grep(count_only) on synthetic logs in /dev/shm.
The first call of a process pays for GPU start-up, the first call of a pattern set for its compile and buffers."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    import hypergrep_amd
    from hypergrep_amd import benchspec, device

    _, needles, hpm = benchspec.c3_spec()
    patterns = ["ERROR", "status=5[0-9]{2}", "timeout after [0-9]+ ms"]
    top = 1 << 30
    text = torch.empty(top + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), top, benchspec.SEED_BASE + 9, needles, hpm)
    torch.cuda.synchronize()
    host = text[:top].cpu().numpy()
    del text
    torch.cuda.empty_cache()
    path = f"/dev/shm/hg_latency_{os.getpid()}.log"
    try:
        for size in (64 << 10, 1 << 20, 16 << 20, 256 << 20, 1 << 30):
            cut = size
            while cut > 0 and host[cut - 1] != 10:  # whole lines
                cut -= 1
            with open(path, "wb") as f:
                f.write(host[:cut].tobytes())
            times = []
            for _ in range(4):
                t0 = time.perf_counter()
                count, rc = hypergrep_amd.grep(path, patterns, count_only=True)
                times.append(time.perf_counter() - t0)
            print(f"{size >> 10:8d} KiB: rc={rc} lines={count:8d}  first {times[0] * 1e3:9.2f} ms, then {min(times[1:]) * 1e3:9.2f} ms  ({cut / (1 << 30) / min(times[1:]):6.2f} GiB/s)", flush=True)
    finally:
        if os.path.exists(path):
            os.unlink(path)


if __name__ == "__main__":
    main()
