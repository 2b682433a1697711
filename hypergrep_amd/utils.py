"""hypergrep-compatible Python API on top of the MI355X scan engine.

The public surface of the reference's hypergrep/utils.py, name for name and default for default:

    Result / CALLBACK_TYPE          utils.py:25-51      batched-match C struct and callback type
    check_compatibility()           utils.py:97-122     compile-only probe, 0 or 4
    configure_libraries()           utils.py:125-144    library path override, ValueError once loaded
    grep()                          utils.py:147-231    (1-based line number, decoded line) tuples + return code
    prepare_patterns()              utils.py:234-289    str -> C arrays, default flags 14, default ids all 0
    scan()                          utils.py:292-358    FFI call on a daemon thread, 130 on Ctrl-C

The native side is ONE shared object, hypergrep_amd/lib/libhyperscanner.so, exporting the reference shim's
`hyperscan` / `check_patterns` ABI (include/hypergrep_amd.h, Face B).  It replaces both libraries of the
reference bundle (libhs + libhyperscanner); gzip comes from the system zlib and zstd from the system
libzstd, so `configure_libraries(libzstd=...)` only preloads an alternative libzstd.

There is no CPU scan path: on a machine without a usable GPU `scan()` returns 3 (HYPERSCANNER_SCRATCH)
and the native library prints the reason.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading
from typing import Callable, Sequence

# Hyperscan's compile flags, the values the reference passes through (utils.py:10-13).
HS_FLAG_CASELESS, HS_FLAG_DOTALL, HS_FLAG_MULTILINE, HS_FLAG_SINGLEMATCH = 1, 2, 4, 8
_GREP_FLAGS = HS_FLAG_DOTALL | HS_FLAG_MULTILINE | HS_FLAG_SINGLEMATCH  # what grep() and the default of scan() use

RC_INVALID_FILE = 101  # grep(): the path is missing or a directory (utils.py:16)
_RC_INTERRUPTED = 130
_JOIN_TIMEOUT_S = 3600


class Result(ctypes.Structure):
    """One match as the shim lays it out (hyperscanner.c:42-46): report id, 0-based line index, the line's bytes."""

    _fields_ = [("id", ctypes.c_uint), ("line_number", ctypes.c_ulonglong), ("line", ctypes.c_char_p)]


# void on_event(Result *batch, int count)  (hyperscanner.c:54)
CALLBACK_TYPE = ctypes.CFUNCTYPE(None, ctypes.POINTER(Result), ctypes.c_int, use_errno=False, use_last_error=False)


class _Libraries:
    """Paths chosen through configure_libraries() and the handles once loaded (loading is lazy: a forked worker must
    initialise the GPU runtime itself)."""

    engine_path = ""
    zstd_path = ""
    engine = None
    zstd = None

    @staticmethod
    def default_engine() -> str:
        return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhyperscanner.so")


def _get_hyperscanner_lib() -> ctypes.CDLL:
    if _Libraries.zstd is None and _Libraries.zstd_path:
        _Libraries.zstd = ctypes.CDLL(_Libraries.zstd_path, mode=ctypes.RTLD_GLOBAL)
    if _Libraries.engine is None:
        path = _Libraries.engine_path or _Libraries.default_engine()
        if not os.path.exists(path):
            raise OSError(
                f"{path}: native library not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(needs hipcc); hypergrep_amd has no pure-Python or CPU fallback."
            )
        _Libraries.engine = ctypes.cdll.LoadLibrary(path)
    return _Libraries.engine


def configure_libraries(libhs: str | None = None, libzstd: str | None = None) -> None:
    """Choose other library files; only possible before the first scan / check loads them.

    libhs: the engine library (default: this package's libhyperscanner.so).  libzstd: a libzstd to preload for .zst input.
    """
    in_use = _Libraries.engine is not None  # the engine opens its libzstd itself, so both are fixed once it is loaded
    for name, wanted, slot in (("libhs", libhs, "engine_path"), ("libzstd", libzstd, "zstd_path")):
        if not wanted:
            continue
        if in_use:
            raise ValueError(f"{name} already loaded, configuration overrides must be called before library usage")
        setattr(_Libraries, slot, wanted)


def _one_per_pattern(kind: str, values: Sequence[int], default: int, count: int) -> list[int]:
    chosen = list(values) if values else [default] * count
    if len(chosen) != count:
        raise ValueError(
            f"Found {len(chosen)} {kind}, expecting {count}. Hyperscan {kind} must be provided for each regex to compile the database."
        )
    return chosen


def prepare_patterns(patterns: list[str], flags: list[int] = (), ids: list[int] = ()) -> tuple[ctypes.Array, ctypes.Array, ctypes.Array]:
    """Patterns, per-pattern flags (default DOTALL | MULTILINE | SINGLEMATCH) and report ids (default all 0) as C arrays.

    ValueError for an empty pattern or when flags / ids are given but not one per pattern.
    """
    count = len(patterns)
    flag_values = _one_per_pattern("flags", flags, _GREP_FLAGS, count)
    id_values = _one_per_pattern("ids", ids, 0, count)
    for pattern in patterns:
        if not pattern:
            raise ValueError(f'Invalid pattern "{pattern}" found. Please provide a valid regex for Intel Hyperscan.')
    return (
        (ctypes.c_char_p * count)(*[pattern.encode() for pattern in patterns]),
        (ctypes.c_uint * count)(*flag_values),
        (ctypes.c_uint * count)(*id_values),
    )


def check_compatibility(patterns: list, flags: list[int] = ()) -> int:
    """Compile the patterns without scanning anything: 0 if the engine accepts them all, else 4."""
    c_patterns, c_flags, c_ids = prepare_patterns(patterns, flags=flags)
    return _get_hyperscanner_lib().check_patterns(c_patterns, c_flags, c_ids, len(c_patterns))


def scan(  # pylint: disable=too-many-arguments
    path: str,
    patterns: list[str],
    callback: Callable,
    flags: list[int] = (),
    ids: list[int] = (),
    buffer_size: int = 262140,
    buffer_count: int = 16,
    max_match_count: int = 0,
) -> int:
    """Scan a plain / gzip / zstd text file; `callback(matches, count)` receives the hits in batches of `buffer_count`.

    The native call runs on a daemon thread so that Ctrl-C reaches Python (return code 130); otherwise the shim's
    return code (0 = fine, 1-7 as in hyperscanner.c:25-33) comes back.
    """
    c_patterns, c_flags, c_ids = prepare_patterns(patterns, flags=flags, ids=ids)
    c_callback = CALLBACK_TYPE(callback)  # referenced until the call is over
    engine = _get_hyperscanner_lib()
    outcome = [0]

    def native_call() -> None:
        outcome[0] = engine.hyperscan(path.encode(), c_patterns, c_flags, c_ids, len(c_patterns), c_callback, buffer_size, buffer_count,
                                      ctypes.c_ulonglong(max_match_count))

    worker = threading.Thread(target=native_call, daemon=True)
    worker.start()
    try:
        worker.join(timeout=_JOIN_TIMEOUT_S)
    except KeyboardInterrupt:
        return _RC_INTERRUPTED
    return outcome[0]


class _GrepSink:
    """on_match for grep(): counts, keeps whole lines, or keeps the matched parts (`re.finditer` of the pattern whose
    id the hit carries — with grep()'s all-zero ids that is the first pattern, as in the reference)."""

    def __init__(self, patterns: list[str], count_only: bool, only_matching: bool, errors: str):
        self.count = 0
        self.rows: list[tuple[int, str]] = []
        self.count_only = count_only
        self.errors = errors
        self.finders = [re.compile(pattern) for pattern in patterns] if only_matching else None

    def __call__(self, matches, count: int) -> None:
        if self.count_only:
            self.count += count
            return
        for hit in (matches[i] for i in range(count)):
            text = hit.line.decode(errors=self.errors)
            if self.finders is None:
                self.rows.append((hit.line_number + 1, text))
            else:
                self.rows.extend((hit.line_number + 1, f"{part.group()}\n") for part in self.finders[hit.id].finditer(text))

    def result(self):
        return self.count if self.count_only else self.rows


def grep(  # pylint: disable=too-many-arguments
    file: str,
    patterns: list[str],
    ignore_case: bool = False,
    count_only: bool = False,
    only_matching: bool = False,
    no_messages: bool = False,
    errors: str = "ignore",
    max_match_count: int = 0,
) -> tuple[int | list[tuple[int, str]], int]:
    """grep for Python: (number of matching lines | [(1-based line number, line)], return code).

    A missing path raises FileNotFoundError and a directory ValueError — or, with `no_messages`, comes back as
    (nothing found, 101).  Invalid regexes raise `re.error` before anything is scanned.
    """
    sink = _GrepSink(patterns, count_only, only_matching, errors)
    if not only_matching:
        for pattern in patterns:  # the reference compiles them for -o in every mode: same early failure for bad syntax
            re.compile(pattern)
    problem = None
    if not os.path.exists(file):
        problem = FileNotFoundError("No such file or directory")
    elif os.path.isdir(file):
        problem = ValueError("is a directory")
    if problem is not None:
        if not no_messages:
            raise problem
        return sink.result(), RC_INVALID_FILE
    flags = _GREP_FLAGS | (HS_FLAG_CASELESS if ignore_case else 0)
    return_code = scan(file, patterns, sink, flags=[flags] * len(patterns), max_match_count=max_match_count)
    return sink.result(), return_code
