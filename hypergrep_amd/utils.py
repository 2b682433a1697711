"""hypergrep-compatible Python API on top of the MI355X scan engine.

Mirrors the reference's public surface (hypergrep/utils.py) name for name, default for default:

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
from typing import Callable

# Flags pulled from hs_compile.h (reference utils.py:10-13).
HS_FLAG_CASELESS = 1
HS_FLAG_DOTALL = 2
HS_FLAG_MULTILINE = 4
HS_FLAG_SINGLEMATCH = 8

# Reference utils.py:16.
RC_INVALID_FILE = 101

__libhs__ = None
__libhs_path__ = ""
__libhyperscanner__ = None
__libzstd__ = None
__libzstd_path__ = ""


class Result(ctypes.Structure):
    """One match: pattern id, 0-based line index, line bytes (reference utils.py:25-40; hyperscanner.c:42-46)."""

    _fields_ = [
        ("id", ctypes.c_uint),
        ("line_number", ctypes.c_ulonglong),
        ("line", ctypes.c_char_p),
    ]


CALLBACK_TYPE = ctypes.CFUNCTYPE(
    None,
    ctypes.POINTER(Result),
    ctypes.c_int,
    use_errno=False,
    use_last_error=False,
)


def _default_library() -> str:
    return os.path.join(os.path.abspath(os.path.dirname(__file__)), "lib", "libhyperscanner.so")


def _get_hyperscanner_lib() -> ctypes.CDLL:
    """Lazily load the native library (lazily so that forked worker processes initialise HIP themselves)."""
    global __libhyperscanner__, __libzstd__, __libhs__  # pylint: disable=global-statement
    if __libzstd__ is None and __libzstd_path__:
        __libzstd__ = ctypes.CDLL(__libzstd_path__, mode=ctypes.RTLD_GLOBAL)
    if __libhyperscanner__ is None:
        path = __libhs_path__ or _default_library()
        if not os.path.exists(path):
            raise OSError(
                f"{path}: native library not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(needs hipcc); hypergrep_amd has no pure-Python or CPU fallback."
            )
        __libhyperscanner__ = ctypes.cdll.LoadLibrary(path)
        __libhs__ = __libhyperscanner__
    return __libhyperscanner__


def check_compatibility(patterns: list, flags: list[int] = ()) -> int:
    """Test pattern compilation without scanning a file; 0 if every pattern compiles, else 4."""
    pattern_array, flags_array, ids_array = prepare_patterns(patterns, flags=flags)
    lib = _get_hyperscanner_lib()
    return lib.check_patterns(pattern_array, flags_array, ids_array, len(pattern_array))


def configure_libraries(libhs: str | None = None, libzstd: str | None = None) -> None:
    """Set the paths to library files; must run before first use (reference utils.py:125-144).

    libhs: path of the engine library (this package's libhyperscanner.so by default).
    libzstd: alternative libzstd to preload for .zst input.
    """
    if libhs:
        if __libhs__:
            raise ValueError("libhs already loaded, configuration overrides must be called before library usage")
        global __libhs_path__  # pylint: disable=global-statement
        __libhs_path__ = libhs
    if libzstd:
        if __libzstd__:
            raise ValueError("libzstd already loaded, configuration overrides must be called before library usage")
        global __libzstd_path__  # pylint: disable=global-statement
        __libzstd_path__ = libzstd


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
    """grep-like helper returning (count | [(1-based line number, line)], return code); reference utils.py:147-231."""
    return_code = 0
    compiled_patterns = [re.compile(pattern) for pattern in patterns]
    results = [] if not count_only else 0

    if not os.path.exists(file):
        return_code = RC_INVALID_FILE
        if not no_messages:
            raise FileNotFoundError("No such file or directory")
    if os.path.isdir(file):
        return_code = RC_INVALID_FILE
        if not no_messages:
            raise ValueError("is a directory")

    if not return_code:

        def _c_callback(matches: list, count: int) -> None:
            nonlocal results
            if count_only:
                results += count
            elif only_matching:
                for index in range(count):
                    match = matches[index]
                    line = match.line.decode(errors=errors)
                    for partial in compiled_patterns[match.id].finditer(line):
                        results.append((match.line_number + 1, f"{partial.group()}\n"))
            else:
                for index in range(count):
                    match = matches[index]
                    results.append((match.line_number + 1, match.line.decode(errors=errors)))

        flags = HS_FLAG_DOTALL | HS_FLAG_MULTILINE | HS_FLAG_SINGLEMATCH
        if ignore_case:
            flags |= HS_FLAG_CASELESS
        return_code = scan(file, patterns, _c_callback, flags=[flags for _ in patterns], max_match_count=max_match_count)

    return results, return_code


def prepare_patterns(
    patterns: list[str],
    flags: list[int] = (),
    ids: list[int] = (),
) -> tuple[ctypes.Array, ctypes.Array, ctypes.Array]:
    """Python patterns / flags / ids -> C arrays (reference utils.py:234-289; same defaults and errors)."""
    if not flags:
        flags = [HS_FLAG_DOTALL | HS_FLAG_MULTILINE | HS_FLAG_SINGLEMATCH for _ in patterns]
    if len(flags) != len(patterns):
        raise ValueError(
            f"Found {len(flags)} flags, expecting {len(patterns)}. Hyperscan flags must be provided for each regex to compile the database."
        )
    if not ids:
        ids = [0 for _ in patterns]
    if len(ids) != len(patterns):
        raise ValueError(
            f"Found {len(ids)} ids, expecting {len(patterns)}. Hyperscan ids must be provided for each regex to compile the database."
        )
    encoded_patterns = []
    for pattern in patterns:
        if not pattern:
            raise ValueError(f'Invalid pattern "{pattern}" found. Please provide a valid regex for Intel Hyperscan.')
        encoded_patterns.append(pattern.encode())
    pattern_array = (ctypes.c_char_p * (len(encoded_patterns)))()
    pattern_array[:] = encoded_patterns
    flags_array = (ctypes.c_uint * (len(flags)))()
    flags_array[:] = [ctypes.c_uint(flag) for flag in flags]
    ids_array = (ctypes.c_uint * (len(ids)))()
    ids_array[:] = [ctypes.c_uint(id_num) for id_num in ids]
    return pattern_array, flags_array, ids_array


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
    """Scan a plain / gzip / zstd text file; `callback(matches, count)` receives batches (reference utils.py:292-358)."""
    pattern_array, flags_array, ids_array = prepare_patterns(patterns, flags=flags, ids=ids)
    callback = CALLBACK_TYPE(callback)
    lib = _get_hyperscanner_lib()
    ret_code = 0

    def _wrapper() -> None:
        nonlocal ret_code
        ret_code = lib.hyperscan(
            path.encode(),
            pattern_array,
            flags_array,
            ids_array,
            len(pattern_array),
            callback,
            buffer_size,
            buffer_count,
            ctypes.c_ulonglong(max_match_count),
        )

    thread = threading.Thread(target=_wrapper, daemon=True)
    thread.start()
    try:
        thread.join(timeout=3600)
    except KeyboardInterrupt:
        ret_code = 130
    return ret_code
