#!/usr/bin/env python3
"""End-to-end (Face B) rate: file on tmpfs -> hyperscan() -> callbacks.  PCIe- and IO-inclusive, never bench.py's `value`.
    python tools/e2e_bench.py [--gib 4] [--workload c3] [--reps 3] [--compress gz|zst]
With --compress the file is written as concatenated gzip members / zstd frames of 64 MiB of text each (compressed by a thread
pool, level 1); rates are in UNCOMPRESSED GiB/s.  HYPERGREP_TRACE=1 in the environment adds the shim's own breakdown."""
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
    ap.add_argument("--compress", default="none", choices=["none", "gz", "zst"])
    args = ap.parse_args()
    import torch

    import hypergrep_amd
    from hypergrep_amd import benchspec, device

    patterns, needles, hpm = {"c1": benchspec.c1_spec, "c2": benchspec.c2_spec, "c3": benchspec.c3_spec, "c5": benchspec.c5_spec}[args.workload]()
    nbytes = int(args.gib * (1 << 30))
    path = os.path.join(args.dir, f"hg_e2e_{os.getpid()}.log" + {"none": "", "gz": ".gz", "zst": ".zst"}[args.compress])
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    device.synth_device(text.data_ptr(), nbytes, benchspec.SEED_BASE + 3, needles, hpm)
    torch.cuda.synchronize()
    def packer():
        if args.compress == "gz":
            import zlib

            def pack(raw: bytes) -> bytes:
                c = zlib.compressobj(1, zlib.DEFLATED, 31)
                return c.compress(raw) + c.flush()
            return pack
        if args.compress == "zst":
            import ctypes

            z = ctypes.CDLL("libzstd.so.1")
            z.ZSTD_compressBound.restype = ctypes.c_size_t
            z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            z.ZSTD_compress.restype = ctypes.c_size_t
            z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]

            def pack(raw: bytes) -> bytes:
                cap = z.ZSTD_compressBound(len(raw))
                buf = ctypes.create_string_buffer(cap)
                n = z.ZSTD_compress(buf, cap, raw, len(raw), 1)
                return buf.raw[:n]
            return pack
        return lambda raw: raw

    pack = packer()
    t_pack = time.perf_counter()
    with open(path, "wb") as f:
        from concurrent.futures import ThreadPoolExecutor

        step = 64 << 20
        with ThreadPoolExecutor(max_workers=max(1, min(15, (os.cpu_count() or 2) - 1))) as pool:  # (zlib / libzstd release the GIL)
            offs = list(range(0, nbytes, step))
            for group in range(0, len(offs), 30):
                raws = [text[off:min(off + step, nbytes)].cpu().numpy().tobytes() for off in offs[group:group + 30]]
                for piece in pool.map(pack, raws):
                    f.write(piece)
    print(f"file: {os.path.getsize(path) / (1 << 30):.2f} GiB on disk for {nbytes / (1 << 30):.2f} GiB of text ({args.compress}), written in {time.perf_counter() - t_pack:.1f} s", flush=True)
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
            print(f"rep {rep}: rc={rc} hits={n[0]} {dt:.3f} s  {nbytes / (1 << 30) / dt:.2f} GiB/s end to end (uncompressed bytes)", flush=True)
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
