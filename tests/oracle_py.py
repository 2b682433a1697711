"""ctypes face of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

DEFAULT_FLAGS = 2 | 4 | 8  # DOTALL | MULTILINE | SINGLEMATCH (hypergrep/utils.py:258)


class OracleResult(ctypes.Structure):
    _fields_ = [("id", ctypes.c_uint), ("line_number", ctypes.c_ulonglong), ("line", ctypes.c_char_p)]


class OracleHit(ctypes.Structure):
    _fields_ = [
        ("line_number", ctypes.c_uint64),
        ("id", ctypes.c_uint32),
        ("to", ctypes.c_uint32),
        ("line_off", ctypes.c_uint64),
        ("line_len", ctypes.c_uint32),
        ("pad", ctypes.c_uint32),
    ]


EVENT_FN = ctypes.CFUNCTYPE(None, ctypes.POINTER(OracleResult), ctypes.c_int)

_lib = None


def build() -> None:
    subprocess.check_call(["make", "-C", ORACLE_DIR, "_build/liboracle.so"], stdout=subprocess.DEVNULL)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_scan_buffer.restype = ctypes.c_int
        _lib.oracle_scan_buffer.argtypes = [
            ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint),
            ctypes.POINTER(ctypes.c_uint), ctypes.c_uint, ctypes.c_int, ctypes.c_ulonglong,
            ctypes.POINTER(ctypes.POINTER(OracleHit)), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint64),
        ]
        _lib.oracle_free.argtypes = [ctypes.c_void_p]
        _lib.oracle_hyperscan.restype = ctypes.c_int
        _lib.oracle_check_patterns.restype = ctypes.c_int
    return _lib


def _arrays(patterns, flags, ids):
    n = len(patterns)
    enc = [p.encode() if isinstance(p, str) else p for p in patterns]
    pa = (ctypes.c_char_p * n)(*enc)
    fa = (ctypes.c_uint * n)(*(flags if flags else [DEFAULT_FLAGS] * n))
    ia = (ctypes.c_uint * n)(*(ids if ids else [0] * n))
    return pa, fa, ia, n


def check_patterns(patterns, flags=None, ids=None) -> int:
    pa, fa, ia, n = _arrays(patterns, flags, ids)
    return lib().oracle_check_patterns(pa, fa, ia, n)


def scan_buffer(data: bytes, patterns, flags=None, ids=None, buffer_size: int = 262140, max_match_count: int = 0):
    """Returns (rc, hits, n_lines); hits = list of (line_number, id, to, line_off, line_len)."""
    pa, fa, ia, n = _arrays(patterns, flags, ids)
    hits = ctypes.POINTER(OracleHit)()
    nh = ctypes.c_size_t(0)
    nl = ctypes.c_uint64(0)
    rc = lib().oracle_scan_buffer(data, len(data), pa, fa, ia, n, buffer_size, max_match_count,
                                  ctypes.byref(hits), ctypes.byref(nh), ctypes.byref(nl))
    out = [(hits[i].line_number, hits[i].id, hits[i].to, hits[i].line_off, hits[i].line_len) for i in range(nh.value)]
    lib().oracle_free(hits)
    return rc, out, nl.value


def scan_file(path: str, patterns, flags=None, ids=None, buffer_size: int = 262140, buffer_count: int = 16,
              max_match_count: int = 0):
    """Returns (rc, rows, batches); rows = list of (line_number, id, line_bytes)."""
    pa, fa, ia, n = _arrays(patterns, flags, ids)
    rows, batches = [], []

    def cb(res, count):
        batches.append(count)
        for i in range(count):
            rows.append((res[i].line_number, res[i].id, res[i].line))

    fn = EVENT_FN(cb)
    rc = lib().oracle_hyperscan(path.encode(), pa, fa, ia, n, fn, ctypes.c_int(buffer_size), ctypes.c_int(buffer_count),
                                ctypes.c_ulonglong(max_match_count))
    return rc, rows, batches
