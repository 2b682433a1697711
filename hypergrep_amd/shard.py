"""Multi-GPU glue for sharded scans: one process per GPU, shards are independent (no data-path collective).

The only exchange after the per-shard scans:
  1. all_gather of (lines, hits) per rank -> exclusive prefix of lines = this shard's global line offset;
  2. hit records (int64 pairs: line_number | id + (to << 32)) sent to rank 0 — a direct gather, every peer on its own link.
Works with any torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" on CPU for the tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def exchange_counts(n_lines: int, n_hits: int, device) -> torch.Tensor:
    """Returns an int64 tensor [world, 2] of every rank's (lines, hits), on the host (blocks until the exchange is done)."""
    return CountExchange(device).start(n_lines, n_hits).result()


class CountExchange:
    """The all_gather of (lines, hits) without a host stall: start() enqueues the collective and an asynchronous copy of its
    result to pinned memory; result() is asked for one step LATER (the gather of step k's hits runs while step k + 1 scans),
    when the copy has long finished.  On CPU tensors (gloo, the tests) everything is synchronous."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._host = None
        self._event = None

    def start(self, n_lines: int, n_hits: int) -> "CountExchange":
        world = dist.get_world_size()
        mine = torch.tensor([n_lines, n_hits], dtype=torch.int64, device=self.device)
        out = torch.zeros((world, 2), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out.view(-1), mine)
        if self.device.type == "cuda":
            self._host = torch.empty((world, 2), dtype=torch.int64, pin_memory=True)
            self._host.copy_(out, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
            self._keep = out
        else:
            self._host = out
        return self

    def result(self) -> torch.Tensor:
        if self._event is not None:
            self._event.synchronize()
        return self._host


def line_offset(totals: torch.Tensor, rank: int) -> int:
    return int(totals[:rank, 0].sum())


def gather_hits(local_hits: torch.Tensor, totals: torch.Tensor, recv_bufs: list | None = None):
    """local_hits: int64 [n, 2] records with GLOBAL line numbers.  Rank 0 returns the list of per-rank tensors
    (its own first); other ranks return None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    ops, bufs = [], []
    if rank == 0:
        for src in range(1, world):
            n = int(totals[src, 1])
            buf = recv_bufs[src - 1][:n] if recv_bufs else torch.empty((n, 2), dtype=torch.int64, device=local_hits.device)
            bufs.append(buf)
            if n:
                ops.append(dist.P2POp(dist.irecv, buf, src))
    elif local_hits.shape[0]:
        ops.append(dist.P2POp(dist.isend, local_hits, 0))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    return [local_hits] + bufs if rank == 0 else None
