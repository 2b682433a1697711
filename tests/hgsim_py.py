"""ctypes face of tests/native/libhgsim.so — TEST-ONLY host harness around the product's compiler and the
scalar device logic in hypergrep_amd/csrc/hg_core.h (see tests/native/hostsim.cpp)."""
from __future__ import annotations

import ctypes
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "native", "hostsim.cpp")
LIB = os.path.join(REPO, "tests", "native", "libhgsim.so")
CSRC = os.path.join(REPO, "hypergrep_amd", "csrc")
DEFAULT_FLAGS = 14


class SimHit(ctypes.Structure):
    _fields_ = [("line_no", ctypes.c_uint64), ("id", ctypes.c_uint32), ("to", ctypes.c_uint32),
                ("start", ctypes.c_uint64), ("len", ctypes.c_uint32), ("pattern", ctypes.c_uint32)]


_lib = None


def build() -> None:
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("hg_compile.cpp", "hg_compile.h", "hg_core.h", "hg_db.h", "hg_post.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return
    # built aside and renamed into place: parallel test workers (pytest -n) may all find the library stale at once
    tmp = f"{LIB}.{os.getpid()}.tmp"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", tmp, SRC, os.path.join(CSRC, "hg_compile.cpp")])
    os.replace(tmp, LIB)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.hgsim_compile.restype = ctypes.c_void_p
        _lib.hgsim_free.argtypes = [ctypes.c_void_p]
        _lib.hgsim_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        _lib.hgsim_pattern_tier.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        _lib.hgsim_pattern_tier.restype = ctypes.c_uint32
        _lib.hgsim_pattern_nodes.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        _lib.hgsim_pattern_nodes.restype = ctypes.c_uint32
        _lib.hgsim_nfa.restype = ctypes.c_size_t
        _lib.hgsim_nfa.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t]
        _lib.hgsim_scan.restype = ctypes.c_long
        _lib.hgsim_scan.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int,
                                    ctypes.POINTER(ctypes.POINTER(SimHit)), ctypes.POINTER(ctypes.c_uint64)]
        _lib.hgsim_free_hits.argtypes = [ctypes.POINTER(SimHit)]
    return _lib


class Db:
    def __init__(self, patterns, flags=None, ids=None):
        n = len(patterns)
        enc = [p.encode() if isinstance(p, str) else p for p in patterns]
        pa = (ctypes.c_char_p * n)(*enc)
        fa = (ctypes.c_uint * n)(*(flags if flags else [DEFAULT_FLAGS] * n))
        ia = (ctypes.c_uint * n)(*(ids if ids else [0] * n))
        err = ctypes.create_string_buffer(256)
        self.h = lib().hgsim_compile(pa, fa, ia, n, err, 256)
        self.error = err.value.decode() if not self.h else None

    def ok(self) -> bool:
        return bool(self.h)

    def info(self) -> dict:
        out = (ctypes.c_uint32 * 6)()
        lib().hgsim_info(self.h, out)
        return dict(zip(("npatterns", "nfactors", "nwindows", "nslow", "fold_mask", "max_nw"), out))

    def selfcheck(self) -> dict:
        """Table invariants (hostsim.cpp::hgsim_selfcheck): violations must be 0."""
        out = (ctypes.c_uint32 * 6)()
        lib().hgsim_selfcheck.restype = ctypes.c_uint32
        bad = lib().hgsim_selfcheck(ctypes.c_void_p(self.h), out)
        return {"violations": bad, "filter_log2": out[0], "wide": out[1], "crowded_slots": out[2], "byte_windows": out[3], "shared_windows": out[4], "table_first": out[5]}

    def tune(self, sample: bytes) -> int:
        lib().hgsim_tune.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        return lib().hgsim_tune(self.h, sample, len(sample))

    def tier(self, i: int) -> int:
        return lib().hgsim_pattern_tier(self.h, i)

    def nfa(self, pattern: int, data: bytes):
        tos = (ctypes.c_uint32 * (len(data) + 2))()
        n = lib().hgsim_nfa(self.h, pattern, data, len(data), tos, len(data) + 2)
        return list(tos[:n])

    def scan(self, data: bytes, buffer_size: int = 262140):
        """Returns (hits, stats); hits = list of (line_no, id, to, start, len)."""
        if not self.h:
            raise RuntimeError(f"database did not compile: {self.error}")
        out = ctypes.POINTER(SimHit)()
        stats = (ctypes.c_uint64 * 5)()
        n = lib().hgsim_scan(self.h, data, len(data), buffer_size, ctypes.byref(out), stats)
        if n < 0:
            raise RuntimeError(f"hgsim_scan rc {n}")
        hits = [(out[i].line_no, out[i].id, out[i].to, out[i].start, out[i].len) for i in range(n)]
        lib().hgsim_free_hits(out)
        return hits, dict(zip(("bitmap_hits", "candidates", "raw_hits", "pieces", "level2_hits"), stats))

    def __del__(self):
        if getattr(self, "h", None):
            lib().hgsim_free(self.h)
            self.h = None
