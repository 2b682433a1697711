#!/usr/bin/env python3
"""Run bench.py against an experiment build of the native library:
    python tools/variant_bench.py build/variants/w6.so --steps 5 ...
(build one with HG_BUILD_DEFINES / HG_BUILD_OUT, see hypergrep_amd/build.py)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import hypergrep_amd  # noqa: E402

hypergrep_amd.configure_libraries(libhs=os.path.abspath(sys.argv[1]))
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
