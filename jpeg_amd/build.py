"""Build jpeg_amd/libjpeg_amd.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m jpeg_amd.build [--force]

-ffp-contract=off is part of the numeric contract: the reference's float32 operations
must not be fused (DESIGN.md "Arithmetic contract").
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libjpeg_amd.so")

SOURCES = ["kernels_stage.hip", "kernels_fused.hip", "kernels_quad.hip", "kernels_encode.hip", "capi.hip", "entropy.cpp", "entropy_encode.cpp"]
HEADERS = ["dct.hpp", "kernels.hpp", "upsample.hpp", "fused_common.hpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-command-line-argument"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(INCLUDE, "jpeg_amd.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not (force or _stale()):
        return LIB
    cc = hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [cc, *FLAGS, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
