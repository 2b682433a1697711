#!/usr/bin/env python3
"""Benchmark of the MI355X line-scan hot path (BASELINE.json metric: GiB/s scanned + matches/s, 256 patterns over a
32 GiB synthetic log per GPU, text resident in HBM).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...     (no WORLD_SIZE in the environment: starts the line above itself, as a child
                                      process, before anything touches the GPU, and relays its output and exit code)

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
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def _self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: one rank per GPU through torch.distributed.run, as a CHILD process
    (a process that has initialised the GPU must never be replaced by another program; this parent has not even imported
    torch).  Rank 0's JSON line reaches stdout through the inherited descriptor; the exit code is the launcher's, i.e.
    non-zero if any rank failed."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


class _DryScanner:
    """--dry-run only: stands in for the device scanner so that the launch, the counts exchange, the hit gather and the JSON
    line can be rehearsed on a box without a GPU (tests/test_bench_launch.py, gloo).  It scans nothing: every step reports the
    same made-up counts and records, and the line says so ("data": "dry-run")."""

    def __init__(self, torch, rank: int, nbytes: int, hits: int):
        self.torch, self.rank, self.nbytes, self.n = torch, rank, nbytes, hits

    def scan(self, _ptr, nbytes, stream=None):  # noqa: ARG002
        from types import SimpleNamespace

        return SimpleNamespace(n_lines=nbytes // 120 + self.rank, n_hits=self.n, n_candidates=self.n, n_raw_hits=self.n, ms_stream=1.0,
                               ms_total=1.0, reruns=0, stream_launches=1, joiner_launches=0, joiner_tiles=0)

    def fill(self, buf, n: int) -> int:
        buf[:n, 0] = self.torch.arange(n, dtype=self.torch.int64) * 7  # shard-local line numbers, ascending
        buf[:n, 1] = self.rank | (5 << 32)
        return n


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
    ap.add_argument("--dry-run", action="store_true", help="no GPU, no scan: rehearse launch + exchange + gather + JSON with made-up counts (CPU tests)")
    ap.add_argument("--dry-hits", type=int, default=1000, help="--dry-run: hit records per rank and step")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(_self_launch(args.gpus))

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
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
        if args.backend == "nccl":
            args.backend = "gloo"
    else:
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
    if dry:
        text, stream = None, None
        text_ptr = 0
    else:
        text = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
        device.synth_device(text.data_ptr(), nbytes, seed, needles, hpm, first_block=rank * blocks_per_shard, device=local_rank)
        torch.cuda.synchronize()
        text_ptr = text.data_ptr()

    def sync() -> None:
        if not dry:
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

    if dry:
        db, sc = None, _DryScanner(torch, rank, nbytes, args.dry_hits)
    else:
        db, sc = make_scanner(not args.no_tune, ids)
        stream = torch.cuda.current_stream().cuda_stream

    # N > 1: the hit gather of step k runs on its own stream WHILE step k + 1 scans (two buffer sets).  The counts exchange of
    # step k is enqueued right after its scan and read one step later (shard.CountExchange): no host stall per step; the
    # gather of the last step is flushed before the timed region ends.  Every buffer (the two hit slots of each rank and rank
    # 0's receive sets) is allocated by prime() BEFORE the timed region, from the counts of an untimed scan.
    comm_stream = torch.cuda.Stream(device=dev) if (world > 1 and not dry) else None
    hit_bufs: list = [None, None]
    recv_sets: list = [[], []]
    slot_free: list = [None, None]  # event: the gather that last used the slot has finished
    pending: list = []              # (slot, hits copied, counts exchange, event: the copy into the slot is done)
    step_no = [0]

    def prime() -> None:
        """Untimed: one scan, its counts exchanged, every buffer of the gather sized for the largest shard + 25 %."""
        st = sc.scan(text_ptr, nbytes, stream=stream)
        totals = shard.exchange_counts(st.n_lines, st.n_hits, dev)
        cap = int(totals[:, 1].max()) * 5 // 4 + 16
        for slot in (0, 1):
            hit_bufs[slot] = torch.empty((cap, 2), dtype=torch.int64, device=dev)
            if rank == 0:
                recv_sets[slot] = [torch.empty((cap, 2), dtype=torch.int64, device=dev) for _ in range(world - 1)]

    def flush_gather():
        slot, n, counts, ready = pending.pop(0)
        totals = counts.result()  # (pinned host memory: no device round trip)
        hit_buf = hit_bufs[slot]
        if rank == 0 and int(totals[1:, 1].max()) > recv_sets[slot][0].shape[0]:  # (a step with more hits than prime() saw + 25 %: never on a fixed text)
            need = int(totals[1:, 1].max())
            recv_sets[slot] = [torch.empty((need + need // 8, 2), dtype=torch.int64, device=dev) for _ in range(world - 1)]
        if dry:
            hit_buf[:n, 0] += shard.line_offset(totals, rank)
            shard.gather_hits(hit_buf[:n], totals, recv_sets[slot])
            return
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            hit_buf[:n, 0] += shard.line_offset(totals, rank)  # shard-local line numbers -> global
            shard.gather_hits(hit_buf[:n], totals, recv_sets[slot])
            slot_free[slot] = comm_stream.record_event()

    def step():
        st = sc.scan(text_ptr, nbytes, stream=stream)
        if world > 1:
            slot = step_no[0] & 1
            step_no[0] += 1
            counts = shard.CountExchange(dev).start(st.n_lines, st.n_hits)
            if pending:
                flush_gather()  # the previous step's hits travel while this step's results are post-processed and the next scan runs
            if hit_bufs[slot].shape[0] < st.n_hits:
                if slot_free[slot] is not None:
                    slot_free[slot].synchronize()
                hit_bufs[slot] = torch.empty((st.n_hits + st.n_hits // 8 + 16, 2), dtype=torch.int64, device=dev)
            if dry:
                pending.append((slot, sc.fill(hit_bufs[slot], st.n_hits), counts, None))
                return st
            main = torch.cuda.current_stream()
            if slot_free[slot] is not None:
                main.wait_event(slot_free[slot])
            n = sc.copy_hits_to(hit_bufs[slot].data_ptr(), st.n_hits, stream=stream)  # (u64 line, u32 id, u32 to) records
            pending.append((slot, n, counts, main.record_event()))
        return st

    def drain():
        while pending:
            flush_gather()

    if world > 1:
        prime()
        step()  # untimed, whatever --warmup says: the first gather sets up the point-to-point channels of the backend
        drain()
    for _ in range(args.warmup):
        step()
    drain()
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    ms_stream = []
    last = None
    for _ in range(args.steps):
        last = step()
        ms_stream.append(last.ms_stream)
    drain()  # (the last step's gather)
    sync()
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
            "metric": f"GiB/s scanned ({len(patterns)} patterns, {args.gib:g} GiB synthetic log per GPU, text resident in HBM)",
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
            "data": "dry-run: NOTHING was scanned, counts and records are made up (launch / exchange rehearsal)" if dry else "synthetic",
            "config": {"workload": f"{args.workload}: {len(patterns)} patterns ({args.ids} ids) over {args.gib:g} GiB synthetic log per GPU",
                       "bytes_per_gpu": nbytes, "patterns": len(patterns), "lines": total_lines, "hits": total_hits,
                       "parallelism": f"shard{world}",
                       "prefilter_windows": "static" if args.no_tune else "tuned as the file API does on a set's first large file: 1 MiB sampled from the first 256 MiB"},
            "matches_per_s": round(total_hits / (elapsed / args.steps), 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "hg_stream_kernel",
                         "kernel_ms": round(stream_ms, 4), "kernel_ms_source": "HIP events around each launch on its stream, this run",
                         "algorithmic_bytes": algo_bytes, "launches_per_step": launches,
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
            if tr["workload"] == args.workload and abs(tr["gib"] - args.gib) < 1e-9 and not dry:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr["source"]
                if tr.get("kernel_ms_rocprof"):
                    # the committed rocprofv3 --kernel-trace --stats average of the same command (profiles/): rocprof's
                    # durations run ~5 % above the in-run events; `frac` is priced with the LONGER one, the events' figure is kept
                    rp = float(tr["kernel_ms_rocprof"])
                    out["roofline"]["kernel_ms_rocprof"] = rp
                    out["roofline"]["kernel_ms_rocprof_source"] = tr.get("kernel_ms_rocprof_source")
                    out["roofline"]["frac_events"] = out["roofline"]["frac"]
                    out["roofline"]["achieved_events"] = out["roofline"]["achieved"]
                    out["roofline"]["achieved"] = round(algo_bytes / (rp * 1e-3) / 1e9, 2)
                    out["roofline"]["frac"] = round(out["roofline"]["achieved"] / HBM_PEAK_GBPS, 4)
        except (OSError, KeyError, ValueError):
            pass
        if world == 1 and not args.no_extra and not dry:
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
        if world == 1 and args.cpu_seconds > 0 and not dry:
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


def _cgroup_cpus() -> float | None:
    """CPU quota of this process's cgroup in cores (v2 cpu.max, v1 cfs quota), None when unlimited or unreadable."""
    try:
        with open("/sys/fs/cgroup/cpu.max", encoding="ascii") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else int(quota) / int(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", encoding="ascii") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us", encoding="ascii") as f:
            period = int(f.read())
        return quota / period if quota > 0 else None
    except (OSError, ValueError):
        return None


# The only figures the reference publishes for this path (README.md:186-198, restated in BASELINE.md §1): end-to-end CPU
# timings of the real Hyperscan 5.4.2 engine on ANOTHER workload and machine.  Context for the port's numbers, not a baseline.
PUBLISHED_REFERENCE = {
    "source": "/root/reference/README.md:186-198 via BASELINE.md section 1 (derived from the plain-text column: ~3 GB, ~17 M lines)",
    "hardware": "2.10GHz Intel x86_64 Processor (model and core count not stated), real Hyperscan 5.4.2, counting only",
    "GiB_per_s": {"1 pattern, 1 thread": 1.1, "10 patterns, 1 thread": 0.74, "~800 patterns, 1 thread": 0.073, "~800 patterns, 4 files on 4 threads": 0.24},
    "note": "different workload (none of BASELINE.json's configs), different machine, file read + Python callback included",
}


def cpu_baseline(text, nbytes, patterns, ids, sc, target_seconds: float) -> dict:
    """The reference's CPU path on the GPU box's host cores, on a bounded sample of the same text: one thread, then
    `usable cores - 1` workers (the reference's parallel_grep model, hypergrep/multiscanner.py:197: ncpu - 1 workers, each
    with its own piece of the input).  Engine: a genuine libhs if one is installed (kind "reference", driven through
    tests/native/hs_call_order.c), else the oracle's Pike VM (kind "port").  The sample is cut into consecutive block
    ranges of about half a second of single-thread work each; workers draw ranges from a queue until `target_seconds`
    have passed (so the leg is bounded in time whatever share of the cores this job really gets), and the GPU hits of
    everything the workers scanned are checked against their results."""
    import threading

    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py

    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_cpus()
    effective = max(1, min(usable, int(quota + 0.999))) if quota else usable
    workers = max(1, effective - 1)
    real = _find_real_libhs()

    def cut_lines(buf: bytes, lo: int, n: int) -> int:
        """End of the last whole line in buf[lo: lo + n] (lo itself if there is none)."""
        e = buf.rfind(b"\n", lo, min(len(buf), lo + n))
        return e + 1 if e >= 0 else lo

    probe_n = min(nbytes, 256 << 10)
    probe = bytes(text[:probe_n].cpu().numpy())
    probe = probe[: cut_lines(probe, 0, probe_n)]
    t0 = time.perf_counter()
    oracle_py.scan_buffer(probe, patterns, ids=ids)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-6)  # single-thread bytes/s
    unit = max(64 << 10, int(rate * 0.5))
    # host copy of the stretch the legs can reach at most (perfect scaling), capped
    reach = int(min(nbytes, 4 << 30, unit * (2 + 2 * (target_seconds / 0.5) * (workers + 1))))
    host = bytes(text[:reach].cpu().numpy())
    ranges, at = [], 0
    while at < len(host):
        e = cut_lines(host, at, unit)
        if e <= at:
            break
        ranges.append((at, e))
        at = e
    results: list = [None] * len(ranges)
    cursor = [0]
    lock = threading.Lock()

    def leg(nthreads: int, seconds: float):
        """Workers draw ranges in order until the deadline; returns (first range, one past the last range, bytes, seconds)."""
        first = cursor[0]
        deadline = time.perf_counter() + seconds

        def work() -> None:
            while time.perf_counter() < deadline:
                with lock:
                    k = cursor[0]
                    if k >= len(ranges):
                        return
                    cursor[0] = k + 1
                lo, hi = ranges[k]
                results[k] = oracle_py.scan_buffer(host[lo:hi], patterns, ids=ids)

        t_start = time.perf_counter()
        pool = [threading.Thread(target=work) for _ in range(nthreads)]  # (the oracle runs outside the GIL)
        for t in pool:
            t.start()
        for t in pool:
            t.join()
        dt = time.perf_counter() - t_start
        last = cursor[0]
        return first, last, (ranges[last - 1][1] - ranges[first][0]) if last > first else 0, dt

    f1, l1, bytes1, dt1 = leg(1, target_seconds / 2)
    fn, ln, bytesn, dtn = leg(workers, target_seconds) if workers > 1 else (l1, l1, 0, 0.0)
    end = ranges[max(ln, l1) - 1][1] if max(ln, l1) > 0 else 0
    # parity: the GPU scans the whole sample in one call; range-local line numbers / offsets are made global
    want, line0 = [], 0
    for (lo, _hi), res in zip(ranges[: max(ln, l1)], results):
        rc, hits, nl = res
        assert rc == 0
        want += [(n + line0, i, to, a + lo, ll) for (n, i, to, a, ll) in hits]
        line0 += nl
    st = sc.scan(text.data_ptr(), end)
    parity = st.n_lines == line0 and sorted(sc.hits()) == sorted(want)
    single = round(bytes1 / (1 << 30) / max(dt1, 1e-9), 6)
    multi = round(bytesn / (1 << 30) / max(dtn, 1e-9), 6) if workers > 1 and bytesn else single
    out = {"value": multi, "unit": "GiB/s", "cores": workers if workers > 1 and bytesn else 1, "kind": "port", "single_thread_value": single,
           "sample": f"the first {end} bytes of rank 0's shard ({line0} lines, {len(want)} hits) in {max(ln, l1)} consecutive block ranges of ~{unit >> 10} KiB: "
                     f"ranges {f1}..{l1 - 1} on one thread in {dt1:.1f} s, ranges {fn}..{ln - 1} drawn from a queue by {workers} workers in {dtn:.1f} s "
                     f"(the reference's ncpu-1 worker model: usable cores {usable}, cgroup quota {quota if quota else 'none'} -> {effective} effective); "
                     "engine = oracle/ Pike VM (Hyperscan is absent from the reference tree and not installed: the HYPERGREP_LIBHS / ldconfig probe found none)",
           "host_cores": os.cpu_count(), "usable_cores": usable, "cgroup_cpu_quota": quota, "workers": workers, "parity_on_sample": bool(parity),
           "published_reference": PUBLISHED_REFERENCE}
    if real:  # a genuine Hyperscan is installed: time it through the shim's call sequence (one process per core, a file each)
        per = max(1, (ln - fn) // max(workers, 1))
        slices = [(ranges[k][0], host[ranges[k][0]: ranges[min(k + per, ln) - 1][1]]) for k in range(fn, ln, per)]
        out.update(_real_libhs_baseline(real, [(0, b"")] + slices, patterns, ids, workers))
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
