"""Buffer-level Python face of the hg_* C ABI (include/hypergrep_amd.h) for text already resident in HBM.

Used by bench.py, the GPU parity tests and multi-GPU shard drivers.  Device memory is handed over as raw
pointers (e.g. `torch.Tensor.data_ptr()`); no torch types cross into the native library.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

from hypergrep_amd import utils

DEFAULT_FLAGS = utils.HS_FLAG_DOTALL | utils.HS_FLAG_MULTILINE | utils.HS_FLAG_SINGLEMATCH


class HgHit(ctypes.Structure):
    _fields_ = [("line_number", ctypes.c_uint64), ("id", ctypes.c_uint32), ("to", ctypes.c_uint32)]


class HgHitAux(ctypes.Structure):
    _fields_ = [("start", ctypes.c_uint64), ("len", ctypes.c_uint32), ("pattern", ctypes.c_uint32)]


class HgScanResult(ctypes.Structure):
    _fields_ = [
        ("n_hits", ctypes.c_uint64), ("n_lines", ctypes.c_uint64), ("n_candidates", ctypes.c_uint64),
        ("n_raw_hits", ctypes.c_uint64), ("d_hits", ctypes.c_void_p), ("d_aux", ctypes.c_void_p),
        ("ms_stream", ctypes.c_float), ("ms_total", ctypes.c_float), ("reruns", ctypes.c_uint32), ("stream_launches", ctypes.c_uint32),
        ("joiner_tiles", ctypes.c_uint64), ("joiner_launches", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
    ]


class HgDbInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("n_patterns", "n_literal_anchored", "n_always_on", "n_factors", "n_windows",
                                               "fold_mask", "max_state_words", "table_bytes", "byte_windows")]


class HgSynthSpec(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("first_block", ctypes.c_uint64), ("hit_per_million", ctypes.c_uint32),
                ("n_needles", ctypes.c_uint32), ("needles", ctypes.c_char_p), ("needle_off", ctypes.POINTER(ctypes.c_uint32))]


SYNTH_BLOCK = 16000

_configured = False


def lib() -> ctypes.CDLL:
    global _configured
    l = utils._get_hyperscanner_lib()  # pylint: disable=protected-access
    if not _configured:
        l.hg_db_compile.restype = ctypes.c_int
        l.hg_db_compile.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint),
                                    ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_size_t]
        l.hg_db_release.argtypes = [ctypes.c_void_p]
        l.hg_db_tune.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        l.hg_db_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(HgDbInfo)]
        l.hg_scanner_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_size_t]
        l.hg_scanner_destroy.argtypes = [ctypes.c_void_p]
        l.hg_scanner_error.restype = ctypes.c_char_p
        l.hg_scanner_error.argtypes = [ctypes.c_void_p]
        l.hg_scan_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p,
                                     ctypes.POINTER(HgScanResult)]
        l.hg_copy_hits.argtypes = [ctypes.c_void_p, ctypes.POINTER(HgHit), ctypes.POINTER(HgHitAux), ctypes.c_uint64]
        l.hg_copy_hits_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        l.hg_synth_device.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(HgSynthSpec), ctypes.c_int, ctypes.c_void_p]
        l.hg_synth_host.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(HgSynthSpec)]
        l.hg_debug_alloc_guarded.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
        l.hg_debug_free_guarded.argtypes = [ctypes.c_void_p]
        l.hg_debug_free_guarded.restype = None
        l.hg_debug_upload.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64]
        l.hg_debug_download.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64]
        l.hg_faceb_next_device.argtypes = [ctypes.c_int]
        l.hg_faceb_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
        l.hg_faceb_stats.restype = None
        _configured = True
    return l


class CompileError(ValueError):
    """A pattern was rejected by the compiler (what check_compatibility reports as 4)."""


class DeviceError(RuntimeError):
    """HIP failure or no usable GPU.  There is no CPU fallback."""


class Database:
    """Compiled pattern set (host side)."""

    def __init__(self, patterns, flags=None, ids=None):
        pa, fa, ia = utils.prepare_patterns(list(patterns), flags=list(flags or ()), ids=list(ids or ()))
        self._h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = lib().hg_db_compile(pa, fa, ia, len(pa), ctypes.byref(self._h), err, 512)
        if rc != 0:
            raise CompileError(err.value.decode(errors="replace"))

    def tune(self, sample: bytes) -> None:
        """Re-select prefilter windows from a text sample (call before creating a Scanner); results never change."""
        if lib().hg_db_tune(self._h, sample, len(sample)) != 0:
            raise CompileError("hg_db_tune failed")

    def info(self) -> dict:
        out = HgDbInfo()
        lib().hg_db_info(self._h, ctypes.byref(out))
        return {n: getattr(out, n) for n, _ in HgDbInfo._fields_}

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib().hg_db_release(self._h)
            self._h = None


@dataclass
class ScanStats:
    n_hits: int
    n_lines: int
    n_candidates: int
    n_raw_hits: int
    ms_stream: float
    ms_total: float
    reruns: int
    stream_launches: int = 1
    joiner_launches: int = 0
    joiner_tiles: int = 0


class Scanner:
    """Database + workspace resident on one GPU."""

    def __init__(self, db: Database, device: int = 0):
        self.db = db
        self._h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = lib().hg_scanner_create(db._h, device, ctypes.byref(self._h), err, 512)
        if rc != 0:
            raise DeviceError(f"hg_scanner_create failed ({rc}): {err.value.decode(errors='replace')}")
        self._last = HgScanResult()

    def scan(self, d_text: int, nbytes: int, buffer_size: int = 262140, line_base: int = 0, stream: int = 0) -> ScanStats:
        res = HgScanResult()
        rc = lib().hg_scan_device(self._h, ctypes.c_void_p(d_text), nbytes, buffer_size, line_base, ctypes.c_void_p(stream), ctypes.byref(res))
        if rc != 0:
            raise DeviceError(f"hg_scan_device failed ({rc}): {lib().hg_scanner_error(self._h).decode(errors='replace')}")
        self._last = res
        return ScanStats(res.n_hits, res.n_lines, res.n_candidates, res.n_raw_hits, res.ms_stream, res.ms_total, res.reruns, res.stream_launches, res.joiner_launches, res.joiner_tiles)

    @property
    def d_hits(self) -> int:
        """Device pointer of the last scan's hit records (16 B each)."""
        return self._last.d_hits or 0

    def hits(self, limit: int | None = None):
        """Last scan's hits as a list of (line_number, id, to, start, len)."""
        n = self._last.n_hits if limit is None else min(limit, self._last.n_hits)
        if not n:
            return []
        hits = (HgHit * n)()
        aux = (HgHitAux * n)()
        rc = lib().hg_copy_hits(self._h, hits, aux, n)
        if rc != 0:
            raise DeviceError(f"hg_copy_hits failed ({rc})")
        return [(hits[i].line_number, hits[i].id, hits[i].to, aux[i].start, aux[i].len) for i in range(n)]

    def copy_hits_to(self, d_dst: int, limit: int, stream: int = 0) -> int:
        """Async device-to-device copy of the last scan's hit records (16 B each); returns the number copied."""
        n = min(limit, self._last.n_hits)
        rc = lib().hg_copy_hits_device(self._h, ctypes.c_void_p(d_dst), n, ctypes.c_void_p(stream))
        if rc != 0:
            raise DeviceError(f"hg_copy_hits_device failed ({rc})")
        return n

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib().hg_scanner_destroy(self._h)
            self._h = None


def _spec(seed: int, first_block: int, hit_per_million: int, needles):
    blob = b"".join(needles)
    offs = [0]
    for n in needles:
        offs.append(offs[-1] + len(n))
    off_arr = (ctypes.c_uint32 * len(offs))(*offs)
    spec = HgSynthSpec(seed, first_block, hit_per_million, len(needles), blob, off_arr)
    return spec, (blob, off_arr)


def synth_device(d_text: int, nbytes: int, seed: int, needles, hit_per_million: int, first_block: int = 0, device: int = 0, stream: int = 0) -> None:
    """Fill nbytes of device memory with the deterministic synthetic log (see csrc/hg_synth.h)."""
    spec, _keep = _spec(seed, first_block, hit_per_million, needles)
    rc = lib().hg_synth_device(ctypes.c_void_p(d_text), nbytes, ctypes.byref(spec), device, ctypes.c_void_p(stream))
    if rc != 0:
        raise DeviceError(f"hg_synth_device failed ({rc})")


def synth_host(nbytes: int, seed: int, needles, hit_per_million: int, first_block: int = 0) -> bytes:
    """The same bytes produced on the host (needs no GPU)."""
    spec, _keep = _spec(seed, first_block, hit_per_million, needles)
    buf = ctypes.create_string_buffer(nbytes)
    rc = lib().hg_synth_host(buf, nbytes, ctypes.byref(spec))
    if rc != 0:
        raise RuntimeError(f"hg_synth_host failed ({rc})")
    return buf.raw


class GuardedArena:
    """Device memory whose END is followed by unmapped address space (hg_debug_alloc_guarded).  place(data) puts the bytes
    so that they end (rounded up to 16) exactly at that edge and returns the device pointer: a kernel that reads past the
    text faults instead of passing silently.  One arena serves many texts — it is mapped once (re-mapping the same
    addresses per text showed stale reads on this stack)."""

    def __init__(self, capacity: int, device: int = 0):
        self.capacity = (max(capacity, 16) + 15) & ~15
        self._ptr = ctypes.c_void_p()
        self._guard = ctypes.c_void_p()
        rc = lib().hg_debug_alloc_guarded(self.capacity, device, ctypes.byref(self._ptr), ctypes.byref(self._guard))
        if rc != 0:
            raise DeviceError(f"hg_debug_alloc_guarded failed ({rc})")

    def place(self, data: bytes) -> int:
        usable = (len(data) + 15) & ~15
        if usable > self.capacity:
            raise ValueError("text larger than the arena")
        ptr = self._ptr.value + self.capacity - usable
        if data and lib().hg_debug_upload(ctypes.c_void_p(ptr), data, len(data)) != 0:
            raise DeviceError("hg_debug_upload failed")
        return ptr

    def tail(self, nbytes: int) -> bytes:
        """The bytes between the end of a text of `nbytes` placed last and the edge (whatever the memory held before)."""
        n = ((nbytes + 15) & ~15) - nbytes
        out = ctypes.create_string_buffer(max(n, 1))
        if n and lib().hg_debug_download(out, ctypes.c_void_p(self._ptr.value + self.capacity - n), n) != 0:
            raise DeviceError("hg_debug_download failed")
        return out.raw[:n]

    def free(self) -> None:
        if getattr(self, "_guard", None):
            lib().hg_debug_free_guarded(self._guard)
            self._guard = None

    def __del__(self):
        if lib is not None:
            self.free()


def faceb_stats() -> dict:
    """Counters of the file API's caches (hg_faceb_stats)."""
    out = (ctypes.c_uint64 * 8)()
    lib().hg_faceb_stats(out)
    names = ("db_cache_hits", "db_cache_misses", "db_cache_entries", "tunes", "contexts_created", "contexts_reused", "contexts_rebound", "contexts_alive")
    return dict(zip(names, (int(v) for v in out)))


def faceb_next_device(ndev: int) -> int:
    return lib().hg_faceb_next_device(ndev)
