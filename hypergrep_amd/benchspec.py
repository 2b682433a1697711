"""Frozen benchmark workloads (BASELINE.json configs, restated concretely per SURVEY.md §8d).

Each spec is (patterns, needles, hit_per_million): the pattern set handed to the engine, the needle strings the
synthetic-log generator (csrc/hg_synth.h) sprinkles into lines, and the per-line needle probability.
Needles are matching examples of the patterns plus ~20 % near-misses (look-alikes that must NOT match), so the
prefilter sees realistic false candidates.  Everything is derived from a fixed seed; nothing is read from disk.
"""
from __future__ import annotations

import random

SEED_BASE = 0x4859504752455000  # "HYPGRE\0\0"

_STEMS = [
    "disk_quota_exceeded", "oom_killer_invoked", "segfault_at", "kernel_panic", "auth_failure", "tls_handshake_failed",
    "replica_lag_high", "checksum_mismatch", "deadlock_detected", "conn_pool_exhausted", "rate_limit_hit", "cert_expired",
    "snapshot_corrupt", "leader_election", "gc_pause_long", "fd_limit_reached", "dns_resolution_failed", "queue_overflow",
    "heartbeat_missed", "schema_violation", "write_stall", "compaction_backlog", "clock_skew", "split_brain",
    "token_revoked", "quota_denied", "io_timeout", "page_fault_storm", "thermal_throttle", "ecc_error",
    "link_flap", "journal_replay",
]
_ALNUM = "abcdefghijklmnopqrstuvwxyz0123456789"
_HEX = "0123456789abcdef"


def _rand(rng: random.Random, alphabet: str, n: int) -> str:
    return "".join(rng.choice(alphabet) for _ in range(n))


def c3_spec(n_literals: int = 192, n_classes: int = 48, n_anchored: int = 16, seed: int = SEED_BASE + 3):
    """256 patterns: 192 literals (8-24 B), 48 class + bounded-repeat, 16 with anchors / \\b / alternation."""
    rng = random.Random(seed)
    patterns: list[str] = []
    needles: list[bytes] = []
    near: list[bytes] = []
    for i in range(n_literals):
        stem = _STEMS[i % len(_STEMS)]
        want = rng.randint(8, 24)
        lit = f"{stem[: max(3, want - 6)]}_{i:03x}{_rand(rng, _ALNUM, 2)}"[:24]
        while len(lit) < 8:
            lit += rng.choice(_ALNUM)
        patterns.append(lit)
        needles.append(lit.encode())
        if i % 4 == 0:  # look-alike differing in the last byte
            near.append((lit[:-1] + ("x" if lit[-1] != "x" else "y")).encode())
    for i in range(n_classes):
        kind = i % 4
        if kind == 0:
            patterns.append(f"user=[a-z0-9_]{{4,12}} status=5{i % 10}[0-9]")
            needles.append(f"user={_rand(rng, _ALNUM, rng.randint(4, 12))} status=5{i % 10}{rng.randint(0, 9)}".encode())
            near.append(f"user={_rand(rng, _ALNUM, 3)} status=5{i % 10}{rng.randint(0, 9)}".encode())
        elif kind == 1:
            patterns.append(f"txn_[a-f0-9]{{8}} aborted_c{i:02d}")
            needles.append(f"txn_{_rand(rng, _HEX, 8)} aborted_c{i:02d}".encode())
            near.append(f"txn_{_rand(rng, _HEX, 7)}g aborted_c{i:02d}".encode())
        elif kind == 2:
            patterns.append(f"retry_budget_{i:02d}=[0-9]{{1,3}}/[0-9]{{2,4}} exhausted")
            needles.append(f"retry_budget_{i:02d}={rng.randint(0, 999)}/{rng.randint(10, 9999)} exhausted".encode())
        else:
            patterns.append(f"blk_[0-9]+_{i:02d} (?:lost|stale|orphaned)")
            needles.append(f"blk_{rng.randint(1, 10**9)}_{i:02d} {rng.choice(['lost', 'stale', 'orphaned'])}".encode())
            near.append(f"blk_{rng.randint(1, 10**9)}_{i:02d} found".encode())
    for i in range(n_anchored):
        kind = i % 4
        if kind == 0:
            patterns.append(f"\\bpanic_code_{i:x}[0-9a-f]{{4}}\\b")
            needles.append(f"panic_code_{i:x}{_rand(rng, _HEX, 4)}".encode())
            near.append(f"panic_code_{i:x}{_rand(rng, _HEX, 5)}".encode())
        elif kind == 1:
            patterns.append(f"(?:fatal_error_aa{i:x}|fatal_error_bb{i:x})_[0-9]+")
            needles.append(f"fatal_error_{rng.choice(['aa', 'bb'])}{i:x}_{rng.randint(0, 9999)}".encode())
        elif kind == 2:
            patterns.append(f"shutdown_reason_{i:x}=[a-z]+$")
            needles.append(f"shutdown_reason_{i:x}={_rand(rng, 'abcdefgh', 6)}".encode())
        else:
            patterns.append(f"(?i)Unhandled_Exception_{i:x}: [a-z]+error")
            needles.append(f"unhandled_exception_{i:x}: {_rand(rng, 'abcdef', 5)}Error".encode())
    all_needles = needles + near
    # ~1 % of lines carry a MATCHING needle
    hit_per_million = int(10000 * len(all_needles) / len(needles))
    return patterns, all_needles, hit_per_million


def c2_spec(seed: int = SEED_BASE + 2):
    """One regex: character class + bounded repeat (BASELINE config 2)."""
    rng = random.Random(seed)
    patterns = ["user=[a-z0-9_]{4,12} status=5[0-9]{2}"]
    needles = [f"user={_rand(rng, _ALNUM, rng.randint(4, 12))} status=5{rng.randint(0, 99):02d}".encode() for _ in range(48)]
    near = [f"user={_rand(rng, _ALNUM, rng.randint(4, 12))} status=4{rng.randint(0, 99):02d}".encode() for _ in range(16)]
    return patterns, needles + near, int(10000 * 64 / 48)


def c5_spec(n: int = 4096, seed: int = SEED_BASE + 5):
    """4096 literals of the form K%04x-%08x, 10 % of lines hit (BASELINE config 5)."""
    rng = random.Random(seed)
    patterns = [f"K{i:04x}-{rng.getrandbits(32):08x}" for i in range(n)]
    return patterns, [p.encode() for p in patterns], 100000


def c1_spec(seed: int = SEED_BASE + 1):
    """One 8-byte literal, 1 % of lines (BASELINE config 1: 1 MiB file through grep())."""
    return ["zq8Lm4Xw"], [b"zq8Lm4Xw"], 10000
