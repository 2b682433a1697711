"""bench.py's launch contract on CPU: `python bench.py --gpus N` with no launcher in the environment starts one rank per GPU
through torch.distributed.run as a child process and relays rank 0's single JSON line (the driver's command form).  The scan
itself needs a GPU, so these runs use --dry-run: made-up counts and records, the real launch / counts exchange / hit gather /
JSON code (gloo instead of RCCL).  The reference's model for the fan-out: hypergrep/multiscanner.py:197-214."""
from __future__ import annotations

import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra: str, timeout: int = 300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", *extra], capture_output=True, text=True, timeout=timeout, env=env, check=False)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    return proc, lines


def test_self_launch_two_ranks_prints_one_json_line():
    proc, lines = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--gib", "0.01", "--workload", "c2")
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1, proc.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["config"]["parallelism"] == "shard2"
    assert out["config"]["hits"] == 2 * 1000  # all ranks' records of one step
    assert "0.01 GiB" in out["metric"] and "1 patterns" in out["metric"]  # the label follows --gib / --workload
    assert out["data"].startswith("dry-run")


def test_single_rank_needs_no_launcher():
    proc, lines = _run("--gpus", "1", "--steps", "2", "--warmup", "0", "--gib", "0.01")
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


def test_config5_sized_gather_through_the_launcher():
    """config 5 sends ~458 MB per peer; here 2.5 M records (40 MB) per rank and step through the same code."""
    proc, lines = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--gib", "0.01", "--workload", "c5", "--dry-hits", "2500000")
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["config"]["hits"] == 5000000 and out["n_gpus"] == 2


def test_a_failing_rank_fails_the_launcher():
    proc, lines = _run("--gpus", "2", "--steps", "1", "--gib", "0.01", "--backend", "no-such-backend")  # every rank dies in init_process_group
    assert proc.returncode != 0 and not lines
