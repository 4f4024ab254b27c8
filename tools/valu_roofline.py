#!/usr/bin/env python3
"""The BINDING roofline of the fused kernels: VALU issue (round 5).

k_quad420 and k_encode_fused are held below the HBM roofline by the vector ALU: every float operation of the reference is
its own instruction (FMA contraction is forbidden, decode.swift:4042-4093 / encode.swift:123-188).  This module turns SQ
counters of a kernel into "issue cycles the VALU needs per SIMD" and divides by the cycles the kernel ran:

    valu_frac = sum over instruction classes (dynamic count x issue cycles per instruction) / SIMDs / cycles elapsed

* dynamic counts: rocprofv3 --pmc, per counter class -- SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _CVT, _INT32, _INT64 and
  "other" = SQ_INSTS_VALU minus those.  Which opcode is tallied where was CALIBRATED with one kernel per opcode
  (tools/probe_valu_classes.hip under the same counters, profiles/r05_probe_valu_classes.txt); the rules are in `counter_class`.
* issue cycles per instruction and SIMD: the same probe, measured in shader cycles with 8 waves per SIMD (`COSTS`, overridable
  with a JSON file): 2 for the full-rate class, 4 for the half-rate class, 8 for transcendentals (a SIMD issues a wave64
  instruction over 2 / 4 / 8 cycles).
* classes that MIX rates (INT32, "other") take the mean cost of the kernel's own instructions of that class, weighted by
  their STATIC frequency in the kernel's disassembly (llvm-objdump of the built library): the one approximation, reported
  as `mixed_share` (how much of the cycle sum it touches) and bracketed by `valu_frac_min` / `valu_frac_max` (every mixed
  instruction at the cheapest / dearest cost of its class).
* cycles elapsed: SQ_BUSY_CYCLES / 32 (the counter is summed over the 8 XCDs x 4 shader engines); with the kernel's
  duration from the kernel trace of the same pass that is the achieved shader clock, reported as `clock_GHz`.

    tools/valu_roofline.py --lib jpeg_amd/libjpeg_amd.so --kernel k_quad420ILi1ELi32ELb1ELb0 <pass-a dir> <pass-b dir>
"""
from __future__ import annotations

import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

PASS_A = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
          "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"]
PASS_B = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"]

# issue cycles per instruction and SIMD (tools/probe_valu_classes.hip, 8 waves per SIMD; profiles/r05_probe_valu_classes.txt)
FULL, HALF, QUARTER = 2.0, 4.0, 8.0
COSTS_DEFAULT = {"full": FULL, "half": HALF, "trans": QUARTER}

FULL_RATE = re.compile(r"^v_(add|sub|subrev|mul|mac|fmac|fmamk|fmaak|fma)_f32(_e32|_e64)?$|^v_(mov_b32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32)(_e32|_e64)?$")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag)?_f(16|32)")


def rate_class(op: str) -> str:
    """full / half / trans -- the probe's three issue rates.  Everything that is not a plain f32 add / sub / mul / fma or one of
    the few full-rate integer operations (v_mov, v_add_u32, v_and / or / xor) issues at half rate on gfx950: conversions,
    v_cvt_pk_*, v_perm_b32, min / max / med3, floor, shifts, three-operand integer ops, SDWA and DPP forms, v_pk_*."""
    if "sdwa" in op or "dpp" in op:
        return "half"
    if TRANS.match(op):
        return "trans"
    return "full" if FULL_RATE.match(op) else "half"


def counter_class(op: str) -> str:
    """The SQ_INSTS_VALU_* counter an opcode is tallied in (calibrated, see the module docstring)."""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if TRANS.match(base):
        return "TRANS_F32"
    if re.match(r"^v_cvt_", base):
        return "CVT"
    if re.match(r"^v_(add|sub|subrev)_f32$", base):
        return "ADD_F32"
    if re.match(r"^v_mul_f32$", base):
        return "MUL_F32"
    if re.match(r"^v_(fma|fmac|fmamk|fmaak|mac|mad)_f32$", base):
        return "FMA_F32"
    if re.match(r"^v_(lshl_add_u64|mad_u64_u32|mad_i64_i32|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_co_u32|addc_co_u32)$", base):
        return "INT64"
    if re.match(r"^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|min|max|med3|add3|perm|bitop3|alignb|cmp|dot|sad|mbcnt|bcnt|ff|lsh)[a-z0-9_]*_(u|i|b)(8|16|24|32)", base):
        return "INT32"
    return "OTHER"


def disassemble(lib: str, kernel: str) -> list[str]:
    """Opcodes of the first kernel whose mangled name contains `kernel`, from the gfx950 code objects bundled in `lib`."""
    tmp = tempfile.mkdtemp(prefix="jpeg_amd_isa_", dir="/tmp")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, capture_output=True, check=True)
        for co in sorted(glob.glob(local + ".*gfx950")):
            text = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=True).stdout
            ops, inside = [], False
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    if inside:
                        return ops
                    inside = kernel in m.group(1)
                    continue
                if inside:
                    t = line.strip().split()
                    if t and re.match(r"^[vs]_|^ds_|^buffer_|^global_|^flat_", t[0]):
                        ops.append(t[0])
            if inside and ops:
                return ops
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    raise RuntimeError(f"kernel {kernel!r} not found in {lib}")


def static_mix(ops: list[str]) -> dict:
    """Per counter class: static instruction count by rate class."""
    mix: dict = {}
    for op in ops:
        if not op.startswith("v_") or re.match(r"^v_(readlane|readfirstlane|writelane|nop)", op):
            continue
        mix.setdefault(counter_class(op), {"full": 0, "half": 0, "trans": 0})[rate_class(op)] += 1
    return mix


def read_counters(dirs, kernel_filter: str) -> tuple[dict, dict]:
    """Mean per dispatch of every counter found under `dirs` for kernels whose name contains `kernel_filter`;
    and mean kernel duration (ns) from the kernel traces there."""
    acc: dict = {}
    dur: list = []
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel_filter in r.get("Kernel_Name", ""):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel_filter in r.get("Kernel_Name", ""):
                    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {"ns": (sum(dur) / len(dur)) if dur else None, "n": len(dur)}


def valu_roofline(counters: dict, mix: dict, duration_ns=None, simds: int = 1024, sq_instances: int = 32, costs=None) -> dict:
    costs = dict(COSTS_DEFAULT, **(costs or {}))
    n = {k.replace("SQ_INSTS_VALU_", ""): v for k, v in counters.items() if k.startswith("SQ_INSTS_VALU_")}
    total = counters["SQ_INSTS_VALU"]
    known = sum(n.get(k, 0.0) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "INT32", "INT64"))
    n["OTHER"] = max(0.0, total - known)
    need = need_min = need_max = mixed = 0.0
    per_class = {}
    for cls, cnt in n.items():
        if cls in ("ADD_F32", "MUL_F32", "FMA_F32"):
            c = lo = hi = costs["full"]
        elif cls == "TRANS_F32":
            c = lo = hi = costs["trans"]
        elif cls == "CVT":
            c = lo = hi = costs["half"]
        else:   # INT32 / INT64 / OTHER: the kernel's own static mix of that class
            m = mix.get(cls, {"full": 0, "half": 1, "trans": 0})
            tot = max(1, m["full"] + m["half"] + m["trans"])
            c = (m["full"] * costs["full"] + m["half"] * costs["half"] + m["trans"] * costs["trans"]) / tot
            lo, hi = costs["full"], costs["half"]
            mixed += cnt * c
        need += cnt * c; need_min += cnt * lo; need_max += cnt * hi
        per_class[cls] = {"insts": round(cnt), "cycles_per_inst": round(c, 3)}
    elapsed = counters["SQ_BUSY_CYCLES"] / sq_instances
    out = {"valu_frac": round(need / simds / elapsed, 4),
           "valu_frac_min": round(need_min / simds / elapsed, 4), "valu_frac_max": round(need_max / simds / elapsed, 4),
           "mixed_share": round(mixed / need, 4),
           "valu_issue_cycles_per_simd": round(need / simds), "cycles_elapsed": round(elapsed),
           "insts_valu": round(total), "insts_salu": round(counters.get("SQ_INSTS_SALU", 0)),
           "insts_lds": round(counters.get("SQ_INSTS_LDS", 0)), "insts_vmem": round(counters.get("SQ_INSTS_VMEM", 0)),
           "waves": round(counters.get("SQ_WAVES", 0)), "classes": per_class,
           "cost_model": {"cycles_per_inst": costs, "simds": simds,
                          "note": "issue cycles per wave64 instruction and SIMD: 2 full-rate (f32 add/sub/mul/fma, v_mov, v_add_u32, v_and/or/xor), "
                                  "4 half-rate (conversions, v_cvt_pk, v_perm, min/max, shifts, 3-operand integer, SDWA/DPP), 8 transcendental "
                                  "(tools/probe_valu_classes.hip); INT32/INT64/other weighted by the kernel's static mix"}}
    if duration_ns:
        out["kernel_us_under_counters"] = round(duration_ns / 1e3, 2)
        out["clock_GHz"] = round(elapsed / duration_ns, 3)
    if counters.get("SQ_WAVE_CYCLES") and counters.get("SQ_WAVES"):
        out["wave_cycles_per_wave"] = round(4 * counters["SQ_WAVE_CYCLES"] / counters["SQ_WAVES"])   # the counter ticks every 4 cycles
    return out


def measure(child_cmd: list, kernel_filter: str, lib: str, kernel_symbol: str, timeout: int = 300, keep_dir=None) -> dict:
    """Two rocprofv3 passes (PASS_A, PASS_B; --kernel-trace only beside --pmc) over `child_cmd` -- the program after `--` must be
    the interpreter / binary itself -- then the model above.  Returns the valu record or {"error": ...}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = keep_dir or tempfile.mkdtemp(prefix="jpeg_amd_valu_", dir="/tmp")
    try:
        dirs = []
        for tag, ctrs in (("a", PASS_A), ("b", PASS_B)):
            d = os.path.join(tmp, tag)
            cmd = [exe, "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", tag, "--", *child_cmd]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            if r.returncode != 0:
                return {"error": f"rocprofv3 pass {tag} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]!r}"}
            dirs.append(d)
        counters, dur = read_counters(dirs, kernel_filter)
        if "SQ_INSTS_VALU" not in counters or "SQ_BUSY_CYCLES" not in counters:
            return {"error": f"no counters for {kernel_filter!r}: {sorted(counters)}"}
        rec = valu_roofline(counters, static_mix(disassemble(lib, kernel_symbol)), dur["ns"])
        rec["dispatches"] = dur["n"]
        rec["source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc (two passes: " + " ".join(c.replace("SQ_", "") for c in PASS_A) + " | " +
                         " ".join(c.replace("SQ_", "") for c in PASS_B) + ") over " + " ".join(os.path.basename(c) for c in child_cmd[1:3]))
        return rec
    except Exception as e:   # a side measurement must never take the bench line with it
        return {"error": repr(e)[:300]}
    finally:
        if keep_dir is None:
            shutil.rmtree(tmp, ignore_errors=True)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "jpeg_amd", "libjpeg_amd.so"))
    ap.add_argument("--kernel", required=True, help="substring of the mangled kernel name (disassembly)")
    ap.add_argument("--filter", default=None, help="substring of the kernel name in the counter files (default: --kernel)")
    ap.add_argument("--static-only", action="store_true")
    ap.add_argument("dirs", nargs="*")
    a = ap.parse_args()
    ops = disassemble(a.lib, a.kernel)
    mix = static_mix(ops)
    if a.static_only or not a.dirs:
        hist: dict = {}
        for op in ops:
            if op.startswith("v_"):
                hist[op] = hist.get(op, 0) + 1
        print(f"{a.kernel}: {len(ops)} instructions, {sum(hist.values())} VALU (static)")
        for op, c in sorted(hist.items(), key=lambda t: -t[1]):
            print(f"   {c:6d}  {op:28s} {counter_class(op):10s} {rate_class(op)}")
        print(json.dumps(mix))
        return
    counters, dur = read_counters(a.dirs, a.filter or a.kernel)
    print(json.dumps(valu_roofline(counters, mix, dur["ns"]), indent=1))


if __name__ == "__main__":
    main()
