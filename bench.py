#!/usr/bin/env python3
"""Benchmark of the MI355X line-scan hot path (BASELINE.json metric: GiB/s scanned + matches/s, 256 patterns over a
32 GiB synthetic log per GPU, text resident in HBM).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole scan pipeline (stream kernel, tile scan, confirm, order + de-duplicate) over this
rank's 32 GiB shard; for N > 1 each rank scans its own shard (weak scaling: files / chunks shard with no data-path
exchange), then the ranks all-gather their (lines, hits) counts over RCCL and send their hit records to rank 0; that gather
runs on its own stream and overlaps the next step's scan (every gather has finished when the timed region ends).
Rank 0 prints ONE JSON line.  `roofline` prices the streaming kernel against HBM; `cpu_baseline` times the oracle (a
CPU port of the reference's per-line path) on a bounded sample of the same text on the GPU box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gib", type=float, default=32.0, help="GiB of text per GPU (the headline workload is 32)")
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--ids", default="distinct", choices=["distinct", "shared"], help="pattern ids 0..n-1 or all 0 (grep() semantics)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--no-tune", action="store_true", help="keep the static window selection (no text sample)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target wall time of each cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the untuned / other-ids legs (experiments)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if os.environ.get("HG_LIB"):  # experiment builds (hypergrep_amd/build.py HG_BUILD_OUT)
        import hypergrep_amd

        hypergrep_amd.configure_libraries(libhs=os.path.abspath(os.environ["HG_LIB"]))
    from hypergrep_amd import benchspec, device, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():  # rehearsal of N ranks on fewer GPUs (never the case under the driver)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend=args.backend)

    spec = {"c1": benchspec.c1_spec, "c2": benchspec.c2_spec, "c3": benchspec.c3_spec, "c5": benchspec.c5_spec}[args.workload]
    patterns, needles, hpm = spec()
    ids = list(range(len(patterns))) if args.ids == "distinct" else None
    seed = benchspec.SEED_BASE + {"c1": 1, "c2": 2, "c3": 3, "c5": 5}[args.workload]

    nbytes = int(args.gib * (1 << 30))
    blocks_per_shard = (nbytes + device.SYNTH_BLOCK - 1) // device.SYNTH_BLOCK
    text = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
    device.synth_device(text.data_ptr(), nbytes, seed, needles, hpm, first_block=rank * blocks_per_shard, device=local_rank)
    torch.cuda.synchronize()

    def file_api_sample() -> bytes:
        """What hyperscan() samples by itself on a pattern set's first large file (hg_shim.hip maybe_tune): four pieces of
        256 KiB spread over the first ingest chunk (256 MiB).  The bench tunes on exactly that, so the timed
        configuration is the shipped one."""
        chunk = min(nbytes, 256 << 20)
        pieces = []
        for i in range(4):
            at = (chunk // 4 * i) & ~15
            pieces.append(bytes(text[at: at + min(256 << 10, chunk - at)].cpu().numpy()))
        return b"".join(pieces)

    def make_scanner(tuned: bool, use_ids):
        d = device.Database(patterns, ids=use_ids)
        if tuned:  # setup, untimed; results never depend on it
            d.tune(file_api_sample())
        return d, device.Scanner(d, local_rank)

    db, sc = make_scanner(not args.no_tune, ids)
    stream = torch.cuda.current_stream().cuda_stream

    # N > 1: the hit gather of step k runs on its own stream WHILE step k + 1 scans (two buffer sets).  The counts exchange of
    # step k is enqueued right after its scan and read one step later (shard.CountExchange): no host stall per step; the
    # gather of the last step is flushed before the timed region ends.
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    hit_bufs: list = [None, None]
    recv_sets: list = [[], []]
    slot_free: list = [None, None]  # event: the gather that last used the slot has finished
    pending: list = []              # (slot, hits copied, counts exchange, event: the copy into the slot is done)
    step_no = [0]

    def flush_gather():
        slot, n, counts, ready = pending.pop(0)
        totals = counts.result()
        hit_buf = hit_bufs[slot]
        need = max(int(totals[:, 1].max()), 1)
        if rank == 0 and (not recv_sets[slot] or recv_sets[slot][0].shape[0] < need):
            recv_sets[slot] = [torch.empty((need + need // 8, 2), dtype=torch.int64, device=dev) for _ in range(world - 1)]
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            hit_buf[:n, 0] += shard.line_offset(totals, rank)  # shard-local line numbers -> global
            shard.gather_hits(hit_buf[:n], totals, recv_sets[slot])
            slot_free[slot] = comm_stream.record_event()

    def step():
        st = sc.scan(text.data_ptr(), nbytes, stream=stream)
        if world > 1:
            slot = step_no[0] & 1
            step_no[0] += 1
            counts = shard.CountExchange(dev).start(st.n_lines, st.n_hits)
            if pending:
                flush_gather()  # the previous step's hits travel while this step's results are post-processed and the next scan runs
            if hit_bufs[slot] is None or hit_bufs[slot].shape[0] < st.n_hits:
                if slot_free[slot] is not None:
                    slot_free[slot].synchronize()
                hit_bufs[slot] = torch.empty((st.n_hits + st.n_hits // 8 + 16, 2), dtype=torch.int64, device=dev)
            main = torch.cuda.current_stream()
            if slot_free[slot] is not None:
                main.wait_event(slot_free[slot])
            n = sc.copy_hits_to(hit_bufs[slot].data_ptr(), st.n_hits, stream=stream)  # (u64 line, u32 id, u32 to) records
            pending.append((slot, n, counts, main.record_event()))
        return st

    def drain():
        while pending:
            flush_gather()

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    ms_stream = []
    last = None
    for _ in range(args.steps):
        last = step()
        ms_stream.append(last.ms_stream)
    drain()  # (the last step's gather)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([last.n_hits, last.n_lines], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        total_hits, total_lines = int(tot[0]), int(tot[1])
    else:
        total_hits, total_lines = last.n_hits, last.n_lines

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        gib_s = (nbytes * world / (1 << 30)) / (elapsed / args.steps)
        # the streaming kernel is launched once per pipeline chunk (8 GiB chunks at 32 GiB): all figures are PER LAUNCH
        launches = max(last.stream_launches, 1)
        stream_ms = sum(ms_stream) / len(ms_stream) / launches
        # 1 B read per text byte + 16 B written per hit.  The text bytes of the tiles that the joiner launches
        # streamed (hg_stream_join_kernel, counted by the kernel) are not these launches' work.
        joined_bytes = last.joiner_tiles * 16384
        algo_bytes = (nbytes - joined_bytes + 16 * last.n_hits) // launches
        achieved = algo_bytes / (stream_ms * 1e-3) / 1e9
        out = {
            "metric": "GiB/s scanned (256 patterns, 32 GiB synthetic log per GPU, text resident in HBM)",
            "value": round(gib_s, 3),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {len(patterns)} patterns ({args.ids} ids) over {args.gib:g} GiB synthetic log per GPU",
                       "bytes_per_gpu": nbytes, "patterns": len(patterns), "lines": total_lines, "hits": total_hits,
                       "parallelism": f"shard{world}",
                       "prefilter_windows": "static" if args.no_tune else "tuned as the file API does on a set's first large file: 1 MiB sampled from the first 256 MiB"},
            "matches_per_s": round(total_hits / (elapsed / args.steps), 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "hg_stream_kernel",
                         "kernel_ms": round(stream_ms, 4), "algorithmic_bytes": algo_bytes, "launches_per_step": launches,
                         "joiner": {"kernel": "hg_stream_join_kernel", "launches_per_step": last.joiner_launches, "text_bytes_per_step": joined_bytes},
                         # the whole launch sequence of a step (SURVEY.md §8d t_kernels) priced the same way: all algorithmic bytes / step time
                         "pipeline_frac": round((nbytes + 16 * last.n_hits) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "pipeline": {"candidates": last.n_candidates, "raw_hits": last.n_raw_hits, "reruns": last.reruns,
                         "ms_total_device": round(last.ms_total, 4)},
        }
        # HBM bytes per launch measured with PMC counters (tools/record_round.sh) for the same workload, if committed
        try:
            with open(os.path.join(REPO, "profiles", "hbm_traffic_latest.json"), encoding="utf-8") as f:
                tr = json.load(f)
            if tr["workload"] == args.workload and abs(tr["gib"] - args.gib) < 1e-9:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr["source"]
        except (OSError, KeyError, ValueError):
            pass
        if world == 1 and not args.no_extra:
            # the same pass in the other configurations a caller can meet (each: 2 warm-up + min(steps, 5) timed passes)
            def leg(tuned: bool, use_ids) -> float:
                _d, s2 = make_scanner(tuned, use_ids)
                for _ in range(2):
                    s2.scan(text.data_ptr(), nbytes, stream=stream)
                torch.cuda.synchronize()
                k = max(1, min(args.steps, 5))
                t1 = time.perf_counter()
                for _ in range(k):
                    s2.scan(text.data_ptr(), nbytes, stream=stream)
                torch.cuda.synchronize()
                return round((nbytes / (1 << 30)) / ((time.perf_counter() - t1) / k), 3)

            if not args.no_tune:
                out["value_untuned"] = leg(False, ids)  # static window selection: files under 32 MiB, hg_* callers that never tune
            other = None if ids is not None else list(range(len(patterns)))
            out["value_shared_ids" if other is None else "value_distinct_ids"] = leg(not args.no_tune, other)  # grep() gives every pattern id 0 (utils.py:264-267)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(text, nbytes, patterns, ids, sc, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _find_real_libhs() -> str | None:
    """A genuine Hyperscan (libhs.so.5) to time instead of the oracle: HYPERGREP_LIBHS, else the loader's search path.
    None on this image (Hyperscan is absent from the reference tree and not installed)."""
    import ctypes
    import ctypes.util

    for cand in (os.environ.get("HYPERGREP_LIBHS"), ctypes.util.find_library("hs")):
        if not cand:
            continue
        try:
            lib = ctypes.CDLL(cand)
        except OSError:
            continue
        if hasattr(lib, "hs_compile_multi") and hasattr(lib, "hs_scan") and not hasattr(lib, "hg_db_compile"):  # (not this repository's Face A)
            return cand
    return None


def cpu_baseline(text, nbytes, patterns, ids, sc, target_seconds: float) -> dict:
    """The reference's CPU path on the GPU box's host cores, on a bounded sample of the same text: one thread, then one
    thread per usable core but one, each on its own block range (the reference's parallel_grep model,
    hypergrep/multiscanner.py:197: ncpu - 1 workers).  Engine: a genuine libhs if one is installed (kind "reference",
    driven through tests/native/hs_call_order.c), else the oracle's Pike VM (kind "port").  The GPU hits of the whole
    sample are checked against the union of the threads' results."""
    import threading

    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py

    def lines_prefix(lo: int, n: int) -> bytes:
        host = bytes(text[lo: lo + n].cpu().numpy())
        return host[: host.rfind(b"\n") + 1]

    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(usable - 1, 15))  # (a one-GPU box shares its host: 16 cores are this job's)
    real = _find_real_libhs()
    probe = lines_prefix(0, min(nbytes, 256 << 10))
    t0 = time.perf_counter()
    oracle_py.scan_buffer(probe, patterns, ids=ids)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-6)
    per_thread = int(min(nbytes // (threads + 1), max(len(probe), rate * target_seconds)))
    # consecutive block ranges, each cut at a line boundary
    slices, at = [], 0
    for _ in range(threads + 1):
        piece = lines_prefix(at, per_thread)
        if not piece:
            break
        slices.append((at, piece))
        at += len(piece)
    results = [None] * len(slices)

    def work(i: int) -> None:
        results[i] = oracle_py.scan_buffer(slices[i][1], patterns, ids=ids)

    t0 = time.perf_counter()
    work(0)
    dt1 = time.perf_counter() - t0
    pool = [threading.Thread(target=work, args=(i,)) for i in range(1, len(slices))]  # (the oracle runs outside the GIL)
    t0 = time.perf_counter()
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    dtn = time.perf_counter() - t0
    multi_bytes = sum(len(p) for _, p in slices[1:])
    # parity: the GPU scans the whole sample in one call; slice-local line numbers / offsets are made global
    want, line0 = [], 0
    for (off, _piece), (rc, hits, nl) in zip(slices, results):
        assert rc == 0
        want += [(ln + line0, i, to, lo + off, ll) for (ln, i, to, lo, ll) in hits]
        line0 += nl
    st = sc.scan(text.data_ptr(), at)
    parity = st.n_lines == line0 and sorted(sc.hits()) == sorted(want)
    single = round(len(slices[0][1]) / (1 << 30) / dt1, 6)
    out = {"value": round(multi_bytes / (1 << 30) / dtn, 6) if pool else single, "unit": "GiB/s",
           "cores": len(pool) if pool else 1, "kind": "port", "single_thread_value": single,
           "sample": f"{len(slices)} consecutive block ranges of rank 0's shard ({at} bytes, {line0} lines, {len(want)} hits): range 0 on one thread in {dt1:.1f} s, "
                     f"ranges 1..{len(slices) - 1} on {len(pool)} threads in {dtn:.1f} s (one range each: the reference's ncpu-1 worker model); "
                     "engine = oracle/ Pike VM (Hyperscan is absent from the reference tree and not installed: the HYPERGREP_LIBHS / ldconfig probe found none)",
           "host_cores": os.cpu_count(), "usable_cores": usable, "parity_on_sample": bool(parity)}
    if real:  # a genuine Hyperscan is installed: time it through the shim's call sequence (one process per core, a file each)
        out.update(_real_libhs_baseline(real, slices, patterns, ids, threads))
    return out


def _real_libhs_baseline(libhs: str, slices, patterns, ids, threads: int) -> dict:
    import subprocess
    import tempfile

    exe = os.path.join(REPO, "tests", "native", "hs_call_order")
    if not os.path.exists(exe):
        subprocess.check_call(["gcc", "-O2", "-o", exe, exe + ".c", "-ldl"])
    args = [str(x) for i, p in enumerate(patterns) for x in (14, (ids[i] if ids else 0), p)]
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        paths = []
        for i, (_off, piece) in enumerate(slices[1: threads + 1]):
            paths.append(os.path.join(tmp, f"r{i}.log"))
            with open(paths[-1], "wb") as f:
                f.write(piece)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([exe, libhs, p, "262140", "--"] + args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for p in paths]
        ok = all(p.wait() == 0 for p in procs)
        dt = time.perf_counter() - t0
    total = sum(len(piece) for _off, piece in slices[1: threads + 1])
    return {"kind": "reference", "value": round(total / (1 << 30) / dt, 6), "cores": len(paths), "reference_engine": libhs, "reference_ok": ok}


if __name__ == "__main__":
    main()
