"""Build recipe for the native library: hipcc (gfx950) for the kernels / engine / C ABI, g++ for the pattern compiler.

The result lands IN-TREE at hypergrep_amd/lib/libhyperscanner.so (the name the reference's loader expects,
hypergrep/utils.py:79) so that it travels with the source snapshot; objects go to build/.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libhyperscanner.so")
# Face A artefact: the same objects linked once more with SONAME libhs.so.5, which is what the reference shim's DT_NEEDED
# asks for: hypergrep.configure_libraries(libhs=<this file>) puts the GPU engine under the reference's
# own prebuilt shim.
LIBHS_NAME = "libhs.so.5"
OBJ = os.path.join(REPO, "build", "obj")
# Experiment builds (tools/variant_bench.py): extra -D flags and another output path, e.g.
#   HG_BUILD_DEFINES="-DHG_STREAM_WAVES=6" HG_BUILD_OUT=build/variants/w6.so python hypergrep_amd/build.py
EXTRA = os.environ.get("HG_BUILD_DEFINES", "").split()
if os.environ.get("HG_BUILD_OUT"):
    LIB = os.path.abspath(os.environ["HG_BUILD_OUT"])
    LIB_DIR = os.path.dirname(LIB)
    OBJ = LIB + ".obj"
    LIBHS_NAME = os.path.basename(LIB) + ".libhs.so.5"

HIP_SOURCES = ["hg_stream.hip", "hg_kernels.hip", "hg_always_on.hip", "hg_huge.hip", "hg_engine.hip", "hg_capi.hip", "hg_shim.hip", "hg_hsface.hip"]
CXX_SOURCES = ["hg_compile.cpp"]
HEADERS = ["hg_mem.h", "hg_db.h", "hg_core.h", "hg_post.h", "hg_engine.h", "hg_compile.h", "hg_synth.h", "hg_confirm_dev.h", "hg_sink_dev.h", "hg_tables_dev.h"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the native library can only be built with the ROCm toolchain")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


KERNEL_SOURCES = ("hg_stream.hip", "hg_kernels.hip", "hg_always_on.hip", "hg_huge.hip")  # their per-kernel resources are recorded


def _compile_kernels(cmd: list[str], table_path: str) -> None:
    """Compile a kernel file with the compiler's per-kernel resource remarks on and keep them next to its object."""
    import json
    import re

    proc = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics"], stderr=subprocess.PIPE, text=True)
    remarks, other = [], []
    for line in proc.stderr.splitlines():
        (remarks if "-Rpass-analysis=kernel-resource-usage" in line else other).append(line)
    if other:
        print("\n".join(other), file=sys.stderr)
    if proc.returncode:
        raise subprocess.CalledProcessError(proc.returncode, cmd)
    table, name = {}, None
    for line in remarks:
        m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
        if not m:
            continue
        key, value = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            name = value
            table[name] = {}
        elif name:
            table[name][key] = int(value) if value.isdigit() else value
    with open(table_path, "w", encoding="utf-8") as f:
        json.dump(table, f, indent=1, sort_keys=True)


def _merge_resources() -> None:
    """lib/kernel_resources.json (tests/test_abi.py checks it) = the kernel files' tables; refuse a dword-aligned stream kernel
    that touches scratch: a by-value argument one field too large once cost 20 % of the pass without failing any test."""
    import json

    table = {}
    for src in KERNEL_SOURCES:
        path = os.path.join(OBJ, src + ".resources.json")
        if os.path.exists(path):
            with open(path, encoding="utf-8") as f:
                table.update(json.load(f))
    with open(os.path.join(LIB_DIR, "kernel_resources.json"), "w", encoding="utf-8") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    for kernel, res in table.items():
        # (up to 16 bytes are tolerated: with the draw loop around the tile loop the compiler folds one or two prologue values —
        # read once before the loop, reloaded after it — into scratch; a spill INSIDE the loop shows up as far more)
        if kernel.startswith("_Z16hg_stream_kernelILi") and "ELb0ELi0E" in kernel and res.get("ScratchSize [bytes/lane]", 0) > 16:
            raise RuntimeError(f"{kernel} uses {res['ScratchSize [bytes/lane]']} bytes of scratch per lane: the streaming hot path must stay in registers")


def build(verbose: bool = False, force: bool = False) -> str:
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    common_deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(REPO, "include", "hypergrep_amd.h")]
    objs, jobs = [], []
    for src in HIP_SOURCES + CXX_SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(OBJ, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + common_deps):
            if src.endswith(".hip"):
                cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-pass-failed"] + EXTRA + ["-c", path, "-o", obj]
            else:
                cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((src, cmd))

    def run(job) -> None:
        src, cmd = job
        if src in KERNEL_SOURCES:
            _compile_kernels(cmd, os.path.join(OBJ, src + ".resources.json"))
        else:
            subprocess.check_call(cmd)

    if jobs:  # the translation units are independent: a few at a time (the stream kernels alone take half a minute)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(run, jobs))
    _merge_resources()
    if force or _stale(LIB, objs + [os.path.join(CSRC, "exports.map")]):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-lz", "-ldl", "-Wl,-Bsymbolic", "-Wl,-soname,libhyperscanner.so",
                                                                                "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    libhs = os.path.join(LIB_DIR, LIBHS_NAME)
    if force or _stale(libhs, objs + [os.path.join(CSRC, "exports.map")]):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", libhs] + objs + ["-lz", "-ldl", "-Wl,-Bsymbolic", "-Wl,-soname,libhs.so.5",
                                                                                  "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
