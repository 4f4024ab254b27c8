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

# ---- cost table: SQ cycles per wave64 instruction and SIMD with 8 waves per SIMD, MEASURED per opcode (tools/probe_valu_classes.hip
#      under rocprofv3 --pmc SQ_BUSY_CYCLES, tools/valu_calibrate.py -> profiles/valu_costs.json).  Opcodes the probe does not hold
#      take the cost of their rate class (the median of the probed members).
COSTS_FILE = os.path.join(ROOT, "profiles", "valu_costs.json")
CLASS_DEFAULT = {"full": 2.0, "half": 4.0, "trans": 8.0, "cndmask": 8.0}

FULL_RATE = re.compile(r"^v_(add|sub|subrev|mul|mac|fmac|fmamk|fmaak|fma)_f32$|^v_(mov_b32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|lshrrev_b32|ashrrev_i32|bitop3_b32)$")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag)?_f(16|32)")


def base_op(op: str) -> str:
    return re.sub(r"_(e32|e64)$", "", op)


def rate_class(op: str) -> str:
    """full / half / trans / cndmask -- the probe's issue-rate groups (profiles/r05_probe_valu_classes.txt).  Full rate: plain f32
    add / sub / mul / fma and a few integer operations (v_mov, v_add_u32, v_and / or / xor, right shifts).  Half rate: everything
    else -- conversions, v_cvt_pk_*, v_perm_b32, min / max / med3, floor, left shifts, three-operand integer ops, compares, SDWA
    and DPP forms, v_pk_*.  v_cndmask_b32 issues far slower than either (its own group)."""
    b = base_op(op)
    if b.endswith("_sdwa") or b.endswith("_dpp"):
        return "half"
    if TRANS.match(b):
        return "trans"
    if b.startswith("v_cndmask"):
        return "cndmask"
    return "full" if FULL_RATE.match(b) else "half"


def counter_class(op: str) -> str:
    """The SQ_INSTS_VALU_* counter an opcode is tallied in -- CALIBRATED (one probe kernel per opcode under the counters):
    ADD_F32: v_add / v_sub_f32 (also DPP); MUL_F32: v_mul_f32; FMA_F32: v_fma / fmac / fmamk; CVT: every v_cvt_* (v_cvt_pk_u8_f32,
    SDWA forms, v_cvt_f32_ubyteN, v_cvt_pk_i16_i32 included); TRANS_F32: v_rcp & co.; INT32: integer ARITHMETIC and compares
    (v_add_u32, v_add3, v_lshl_add, v_mul_lo, v_mul / mad_u24, v_ashrrev, v_min / max_i32, v_cmp_*, v_dot4); INT64: 64-bit
    integer ops.  Tallied in NO class ("OTHER" = SQ_INSTS_VALU minus the classes): v_mov, v_cndmask, v_perm, logical ops and
    plain shifts (and / or / xor / bfi / and_or / bitop3 / lshlrev / lshrrev / lshl_or), v_floor / rndne / min / max / med3_f32."""
    b = re.sub(r"_(sdwa|dpp)$", "", base_op(op))
    if TRANS.match(b):
        return "TRANS_F32"
    if b.startswith("v_cvt_"):
        return "CVT"
    if re.match(r"^v_(add|sub|subrev)_f32$", b):
        return "ADD_F32"
    if b == "v_mul_f32":
        return "MUL_F32"
    if re.match(r"^v_(fma|fmac|fmamk|fmaak|mac|mad)_f32$", b):
        return "FMA_F32"
    if re.match(r"^v_(lshl_add_u64|mad_u64_u32|mad_i64_i32|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_co_u32|addc_co_u32|mov_b64)$", b):
        return "INT64"
    if re.match(r"^v_(add|sub|subrev|add3|lshl_add|add_lshl|mul_lo|mul_hi|mul|mad|ashrrev|min|max|med3|dot4|dot2|sad|mbcnt|bcnt)_(u|i)(8|16|24|32)(_(u|i)(8|16|24|32))?$", b) or b.startswith("v_cmp"):
        return "INT32"
    return "OTHER"


def load_costs(path=None, column="w8") -> dict:
    """column: "w8" -- the probe with 8 waves per SIMD, the best the SIMD does: the roofline -- or "w4" / "w3": what it reaches
    with that many resident waves (a kernel held to 3 or 4 waves per SIMD by its LDS / registers cannot issue faster)."""
    path = path or COSTS_FILE
    ops = {}
    if os.path.exists(path):
        ops = {k: v[column] for k, v in json.load(open(path))["ops"].items() if v.get(column)}
    cls = dict(CLASS_DEFAULT)
    for c in ("full", "half", "trans", "cndmask"):
        members = sorted(v for k, v in ops.items() if rate_class(k) == c)
        if members:
            cls[c] = members[len(members) // 2]
    return {"ops": ops, "classes": cls, "column": column, "file": os.path.relpath(path, ROOT) if os.path.exists(path) else None}


def op_cost(op: str, costs: dict) -> float:
    b = base_op(op)
    if b in costs["ops"]:
        return costs["ops"][b]
    if b.endswith("_sdwa") and "v_cvt_f32_i32_sdwa" in costs["ops"]:
        return costs["ops"]["v_cvt_f32_i32_sdwa"]
    if b.endswith("_dpp") and "v_add_f32_dpp" in costs["ops"]:
        return costs["ops"]["v_add_f32_dpp"]
    return costs["classes"][rate_class(op)]


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


def static_mix(ops: list[str], costs=None) -> dict:
    """Per counter class: static instruction count and the sum / extremes of the per-opcode issue costs.  v_cndmask_b32 is left
    out of the class means (it issues ~10 x slower than anything else -- 22.7 cycles -- and sits in the kernels' edge paths:
    weighting it by its STATIC frequency would charge the hot path for code it does not run); its static count is reported
    and valu_frac_max prices the whole class at it."""
    costs = costs or load_costs()
    mix: dict = {}
    for op in ops:
        if not op.startswith("v_") or re.match(r"^v_(readlane|readfirstlane|writelane|nop)", op):
            continue
        c = op_cost(op, costs)
        m = mix.setdefault(counter_class(op), {"n": 0, "cycles": 0.0, "min": c, "max": c, "cndmask": 0})
        m["max"] = max(m["max"], c)
        if rate_class(op) == "cndmask":
            m["cndmask"] += 1
            continue
        m["n"] += 1; m["cycles"] += c; m["min"] = min(m["min"], c)
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


def _need(counters: dict, mix: dict, costs: dict):
    n = {k.replace("SQ_INSTS_VALU_", ""): v for k, v in counters.items() if k.startswith("SQ_INSTS_VALU_")}
    total = counters["SQ_INSTS_VALU"]
    known = sum(n.get(k, 0.0) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "INT32", "INT64"))
    n["OTHER"] = max(0.0, total - known)
    need = need_min = need_max = mixed = 0.0
    per_class = {}
    for cls, cnt in n.items():
        m = mix.get(cls)
        if m and m["n"]:
            c, lo, hi = m["cycles"] / m["n"], min(m["min"], m["cycles"] / m["n"]), m["max"]
        else:   # a class the kernel's disassembly does not hold (stray counts): half rate
            c = lo = hi = costs["classes"]["half"]
        if hi - lo > 0.25 * c:
            mixed += cnt * c
        need += cnt * c; need_min += cnt * lo; need_max += cnt * hi
        per_class[cls] = {"insts": round(cnt), "cycles_per_inst": round(c, 3)}
    return need, need_min, need_max, mixed, per_class


def valu_roofline(counters: dict, ops: list, duration_ns=None, waves_per_simd=None, simds: int = 1024, sq_instances: int = 32) -> dict:
    """counters: per-dispatch means; ops: the kernel's disassembly (opcodes); waves_per_simd: the kernel's occupancy (3 | 4),
    for the second figure -- the fraction of what the SIMD issues with THAT many resident waves."""
    costs = load_costs()
    need, need_min, need_max, mixed, per_class = _need(counters, static_mix(ops, costs), costs)
    elapsed = counters["SQ_BUSY_CYCLES"] / sq_instances
    total = counters["SQ_INSTS_VALU"]
    out = {"valu_frac": round(need / simds / elapsed, 4),
           "valu_frac_min": round(need_min / simds / elapsed, 4), "valu_frac_max": round(need_max / simds / elapsed, 4),
           "mixed_share": round(mixed / need, 4),
           "valu_issue_cycles_per_simd": round(need / simds), "cycles_elapsed": round(elapsed),
           "insts_valu": round(total), "insts_salu": round(counters.get("SQ_INSTS_SALU", 0)),
           "insts_lds": round(counters.get("SQ_INSTS_LDS", 0)), "insts_vmem": round(counters.get("SQ_INSTS_VMEM", 0)),
           "waves": round(counters.get("SQ_WAVES", 0)), "classes": per_class,
           "static_cndmask": sum(m.get("cndmask", 0) for m in static_mix(ops, costs).values())}
    if waves_per_simd in (3, 4):
        c2 = load_costs(column=f"w{waves_per_simd}")
        need2 = _need(counters, static_mix(ops, c2), c2)[0]
        out["waves_per_simd"] = waves_per_simd
        out["valu_frac_at_occupancy"] = round(need2 / simds / elapsed, 4)
        out["occupancy_class_cycles"] = {k: round(v, 3) for k, v in c2["classes"].items()}
    out["cost_model"] = {"class_cycles": {k: round(v, 3) for k, v in costs["classes"].items()}, "per_opcode_table": costs["file"], "simds": simds,
                         "note": "valu_frac = VALU issue cycles needed per SIMD / cycles elapsed (SQ_BUSY_CYCLES / 32).  Needed = dynamic "
                                 "instruction counts per SQ counter class x issue cycles per wave64 instruction, MEASURED per opcode in SQ cycles "
                                 "with 8 waves per SIMD (tools/probe_valu_classes.hip + valu_calibrate.py: 2.3 full rate, 4.1-4.2 half rate, 8.1 "
                                 "transcendental); classes that mix rates take the mean over the kernel's own instructions of the class (static "
                                 "mix of its disassembly, v_cndmask left out), _min / _max price a class at its cheapest / dearest opcode. "
                                 "valu_frac_at_occupancy: the same against the issue rates the probe reaches with the kernel's own number of "
                                 "resident waves per SIMD"}
    if duration_ns:
        out["kernel_us_under_counters"] = round(duration_ns / 1e3, 2)
        out["clock_GHz"] = round(elapsed / duration_ns, 3)
    if counters.get("SQ_WAVE_CYCLES") and counters.get("SQ_WAVES"):
        out["wave_cycles_per_wave"] = round(4 * counters["SQ_WAVE_CYCLES"] / counters["SQ_WAVES"])   # the counter ticks every 4 cycles
    return out


def measure(child_cmd: list, kernel_filter: str, lib: str, kernel_symbol: str, waves_per_simd=None, timeout: int = 300, keep_dir=None) -> dict:
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
        rec = valu_roofline(counters, disassemble(lib, kernel_symbol), dur["ns"], waves_per_simd)
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
    ap.add_argument("--waves", type=int, default=None, help="the kernel's waves per SIMD (3 | 4): adds valu_frac_at_occupancy")
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
            print(f"   {c:6d}  {op:28s} {counter_class(op):10s} {rate_class(op):8s} {op_cost(op, load_costs()):6.2f}")
        print(json.dumps(mix))
        return
    counters, dur = read_counters(a.dirs, a.filter or a.kernel)
    print(json.dumps(valu_roofline(counters, ops, dur["ns"], a.waves), indent=1))


if __name__ == "__main__":
    main()
