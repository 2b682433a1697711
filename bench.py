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
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

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

    db = device.Database(patterns, ids=ids)
    if not args.no_tune:  # prefilter windows chosen from the first 4 MiB of this rank's text (setup, untimed; results unaffected)
        db.tune(bytes(text[: min(nbytes, 4 << 20)].cpu().numpy()))
    sc = device.Scanner(db, local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    # N > 1: the hit gather of step k runs on its own stream and overlaps the scan of step k + 1 (two buffer sets);
    # the final synchronize of the timed region waits for every gather
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    hit_bufs: list = [None, None]
    recv_sets: list = [[], []]
    slot_free: list = [None, None]  # event: the gather that last used the slot has finished
    step_no = [0]

    def step():
        st = sc.scan(text.data_ptr(), nbytes, stream=stream)
        if world > 1:
            slot = step_no[0] & 1
            step_no[0] += 1
            totals = shard.exchange_counts(st.n_lines, st.n_hits, dev)
            # hit records (u64 line, u32 id, u32 to) -> a tensor, shard-local line numbers made global, sent to rank 0
            need = max(int(totals[:, 1].max()), 1)
            if hit_bufs[slot] is None or hit_bufs[slot].shape[0] < need:
                if slot_free[slot] is not None:
                    slot_free[slot].synchronize()
                hit_bufs[slot] = torch.empty((need + need // 8, 2), dtype=torch.int64, device=dev)
                recv_sets[slot] = [torch.empty_like(hit_bufs[slot]) for _ in range(world - 1)] if rank == 0 else []
            main = torch.cuda.current_stream()
            if slot_free[slot] is not None:
                main.wait_event(slot_free[slot])
            hit_buf = hit_bufs[slot]
            n = sc.copy_hits_to(hit_buf.data_ptr(), st.n_hits, stream=stream)
            hit_buf[:n, 0] += shard.line_offset(totals, rank)
            ready = main.record_event()
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ready)
                shard.gather_hits(hit_buf[:n], totals, recv_sets[slot])
                slot_free[slot] = comm_stream.record_event()
        return st

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    ms_stream = []
    last = None
    for _ in range(args.steps):
        last = step()
        ms_stream.append(last.ms_stream)
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
        algo_bytes = (nbytes + 16 * last.n_hits) // launches  # 1 B read per text byte + 16 B written per hit
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
                       "parallelism": f"shard{world}", "prefilter_windows": "static" if args.no_tune else "tuned on the first 4 MiB of the text"},
            "matches_per_s": round(total_hits / (elapsed / args.steps), 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "hg_stream_kernel",
                         "kernel_ms": round(stream_ms, 4), "algorithmic_bytes": algo_bytes, "launches_per_step": launches},
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
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(text, nbytes, patterns, ids, sc, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(text, nbytes, patterns, ids, sc, target_seconds: float) -> dict:
    """Time the oracle (CPU port of the reference's per-line path) on a bounded prefix of the same text and check
    the GPU hits of that prefix against it."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py

    def prefix(n):
        host = bytes(text[:n].cpu().numpy())
        cut = host.rfind(b"\n") + 1
        return host[:cut]

    probe = prefix(min(nbytes, 256 << 10))
    t0 = time.perf_counter()
    oracle_py.scan_buffer(probe, patterns, ids=ids)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-6)
    sample = prefix(int(min(nbytes, max(len(probe), rate * target_seconds))))
    t0 = time.perf_counter()
    rc, want, nlines = oracle_py.scan_buffer(sample, patterns, ids=ids)
    dt = time.perf_counter() - t0
    st = sc.scan(text.data_ptr(), len(sample))
    parity = rc == 0 and st.n_lines == nlines and sorted(sc.hits()) == sorted(want)
    return {"value": round(len(sample) / (1 << 30) / dt, 6), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": f"first {len(sample)} bytes of rank 0's shard, {nlines} lines, {len(want)} hits, {dt:.1f} s single thread; "
                      "Hyperscan itself is unavailable (absent from the reference tree), engine = oracle/ Pike VM",
            "host_cores": os.cpu_count(), "parity_on_sample": bool(parity)}


if __name__ == "__main__":
    main()
