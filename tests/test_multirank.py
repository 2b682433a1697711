"""World-size-2 test of the shard exchange (hypergrep_amd/shard.py) on CPU with gloo: per-shard oracle scans,
all_gather of counts, global line numbers, hit gather to rank 0 == scan of the concatenated text."""
from __future__ import annotations

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_path: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py
    from hypergrep_amd import benchspec, device, shard

    patterns, needles, hpm = benchspec.c3_spec()
    nblocks = 6
    nbytes = nblocks * device.SYNTH_BLOCK
    text = device.synth_host(nbytes, 99, needles, hpm * 10, first_block=rank * nblocks)
    ids = list(range(len(patterns)))
    rc, hits, nlines = oracle_py.scan_buffer(text, patterns, ids=ids)  # stands in for this rank's GPU scan
    assert rc == 0
    totals = shard.exchange_counts(nlines, len(hits), "cpu")
    off = shard.line_offset(totals, rank)
    packed = torch.tensor([[h[0] + off, h[1] | (h[2] << 32)] for h in hits], dtype=torch.int64).reshape(-1, 2)
    gathered = shard.gather_hits(packed, totals)
    if rank == 0:
        whole = b"".join(device.synth_host(nbytes, 99, needles, hpm * 10, first_block=r * nblocks) for r in range(world))
        rc, want, total_lines = oracle_py.scan_buffer(whole, patterns, ids=ids)
        got = torch.cat(gathered).tolist()
        assert int(totals[:, 0].sum()) == total_lines
        assert got == [[h[0], h[1] | (h[2] << 32)] for h in want]
        with open(out_path, "w", encoding="utf-8") as f:
            f.write(f"ok {len(got)}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_exchange(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    status = out.read_text()
    assert status.startswith("ok ") and int(status.split()[1]) > 20


def _volume_worker(rank: int, world: int, port: int, out_path: str, n: int) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hypergrep_amd import shard

    # shard-local records (line, id | to << 32); every rank a different count, as real shards have
    mine = n + 1000 * rank
    local = torch.empty((mine, 2), dtype=torch.int64)
    local[:, 0] = torch.arange(mine, dtype=torch.int64) * 3
    local[:, 1] = (torch.arange(mine, dtype=torch.int64) % 4096) | (torch.full((mine,), 7 + rank, dtype=torch.int64) << 32)
    lines = 3 * mine + 5
    counts = shard.CountExchange("cpu").start(lines, mine)
    totals = counts.result()
    local[:, 0] += shard.line_offset(totals, rank)
    recv = [torch.empty((int(totals[r, 1]) + 17, 2), dtype=torch.int64) for r in range(1, world)] if rank == 0 else None  # (pre-sized, as bench.py's prime())
    got = shard.gather_hits(local, totals, recv)
    if rank == 0:
        assert [int(t.shape[0]) for t in got] == [n + 1000 * r for r in range(world)]
        whole = torch.cat(got)
        assert bool((whole[1:, 0] >= whole[:-1, 0]).all()), "global line numbers must ascend across shards"
        base = 0
        for r in range(world):
            m = n + 1000 * r
            part = got[r]
            assert int(part[0, 0]) == base and int(part[-1, 0]) == base + 3 * (m - 1)
            assert bool(((part[:, 1] >> 32) == 7 + r).all())
            base += 3 * m + 5
        with open(out_path, "w", encoding="utf-8") as f:
            f.write(f"ok {whole.shape[0]}")
    dist.barrier()
    dist.destroy_process_group()


def test_config5_volume_gather(tmp_path):
    """SURVEY.md §8(e) sizes config 5's gather at ~458 MB per peer; 2.2 M records (35 MB) per rank through the same calls."""
    out = tmp_path / "result.txt"
    n = 2_200_000
    mp.spawn(_volume_worker, args=(2, _free_port(), str(out), n), nprocs=2, join=True)
    status = out.read_text()
    assert status == f"ok {2 * n + 1000}"
