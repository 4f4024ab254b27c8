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

SOURCES = ["kernels_stage.hip", "kernels_fused.hip", "kernels_quad.hip", "kernels_encode.hip", "kernels_generic.hip", "capi.hip", "entropy.cpp", "entropy_encode.cpp"]
HEADERS = ["dct.hpp", "kernels.hpp", "upsample.hpp", "fused_common.hpp", "quantise.hpp", "worker_pool.hpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-command-line-argument"]
# Per-source flags.  The transform kernels are compiled WITHOUT the SLP vectoriser: it turns the float arithmetic of the 8-point
# transforms into v_pk_*_f32 -- which issue at half rate (no gain per element), want their literal constants in registers and their
# operands in aligned register pairs (459 v_mov_b32 in k_generic_fused<64, 3>).  Without it (round 6, one box, profiles/r06_no_slp.txt):
# k_generic_fused 135 instead of 160 VGPRs and 5-10 % faster, k_idct_plane 96 instead of 164 VGPRs (100 k blocks 11.8 -> 11.3 us),
# k_encode_fused 4096 x 4096 4:2:0 24.3-25.4 -> 23.4-23.6 us.  Same IEEE operations either way: results are bit-identical.
# (k_quad420 / k_luma_fused pin their arithmetic with empty asm statements and hold no packed operations either way.)
_NO_SLP = ["-fno-slp-vectorize"]
EXTRA_FLAGS = {"kernels_generic.hip": _NO_SLP, "kernels_encode.hip": _NO_SLP, "kernels_stage.hip": _NO_SLP}


# kernels that must not touch scratch memory: source -> mangled-name fragment.  The strip walks count their own VM operations
# (a compiler-placed spill would break the count): for them a spill is a build ERROR.  For the encode and generic kernels a spill is
# a performance bug -- a reload waits with vmcnt(0) for every store in flight -- that has crept in before (round 4: the byte-tail
# encode variants): it is reported, and an error only under JPEG_AMD_STRICT_SPILL=1 (a register-allocation change in a ROCm update
# must not leave a user without a library).
NO_SCRATCH = {"kernels_quad.hip": "k_quad420", "kernels_fused.hip": "k_luma_fused"}
WARN_SCRATCH = {"kernels_encode.hip": "k_encode_fused", "kernels_generic.hip": "k_generic_fused"}


def check_no_scratch(src: str, remarks: str, fragment: str, strict: bool = True) -> None:
    """Parse hipcc's kernel-resource-usage remarks: every kernel whose name contains `fragment` must report
    ScratchSize 0 and no spilled registers."""
    cur, seen, bad = None, 0, []
    for line in remarks.splitlines():
        if "Function Name:" in line:
            cur = line.split("Function Name:")[1].split("[-R")[0].strip()
            seen += fragment in cur
        elif cur and fragment in cur:
            for key in ("ScratchSize [bytes/lane]:", "VGPRs Spill:"):
                if key in line and int(line.split(key)[1].split("[-R")[0].strip()) != 0:
                    bad.append(f"{cur}: {key} {line.split(key)[1].split('[-R')[0].strip()}")
    if seen == 0:
        raise RuntimeError(f"{src}: no resource remarks for {fragment} (the spill gate cannot see the kernels)")
    if bad and strict:
        raise RuntimeError(f"{src}: kernels of the spill gate (NO_SCRATCH) must not spill:\n  " + "\n  ".join(bad))
    if bad:
        sys.stderr.write(f"jpeg_amd.build: warning: {src}: spilling kernels (a performance bug, not an error):\n  " + "\n  ".join(bad) + "\n")


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


def sync_swift_header() -> None:
    """swift/Sources/CJPEGAMD/jpeg_amd.h must be include/jpeg_amd.h (the module map needs the header inside the C target's
    directory); tests/test_abi_cpu.py checks that the two are identical.  In this repository it is a tracked SYMLINK to the
    header and is left alone; where a checkout has a regular file there (no symlink support), the explicit build
    (`python -m jpeg_amd.build`) refreshes the copy -- never a library load, which may run on a read-only checkout or as
    several ranks at once -- and it writes atomically."""
    src = os.path.join(INCLUDE, "jpeg_amd.h")
    dst = os.path.join(os.path.dirname(HERE), "swift", "Sources", "CJPEGAMD", "jpeg_amd.h")
    if not os.path.isdir(os.path.dirname(dst)):
        return
    if os.path.islink(dst) and os.path.realpath(dst) == os.path.realpath(src):
        return
    data = open(src, "rb").read()
    try:
        if os.path.exists(dst) and open(dst, "rb").read() == data:
            return
        tmp = f"{dst}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, dst)
    except OSError as e:
        sys.stderr.write(f"jpeg_amd.build: could not refresh {dst}: {e}\n")


def build(force: bool = False, verbose: bool = False) -> str:
    if not (force or _stale()):
        return LIB
    cc = hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        if src in NO_SCRATCH or src in WARN_SCRATCH:
            # these kernels count their own VM operations (`s_waitcnt vmcnt(16)` behind the coefficient DMA): a compiler-
            # generated scratch store or reload in that window would break the count, so a spill is a build error
            proc = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
            remarks = [l for l in proc.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l]
            if remarks:
                sys.stderr.write("\n".join(remarks) + "\n")
            if proc.returncode != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
            if src in NO_SCRATCH:
                check_no_scratch(src, proc.stderr, NO_SCRATCH[src])
            else:
                check_no_scratch(src, proc.stderr, WARN_SCRATCH[src], strict=os.environ.get("JPEG_AMD_STRICT_SPILL") == "1")
        else:
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    sync_swift_header()
    print(build(force="--force" in sys.argv, verbose=True))
