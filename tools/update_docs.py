#!/usr/bin/env python3
"""Rewrite the measured numbers of DESIGN.md section 6 and of README.md from the committed round profile:
    python tools/update_docs.py [tag]        (default r06: profiles/<tag>_bench.json, profiles/<tag>_traffic.json)
Only the text between the `<!-- numbers:begin -->` / `<!-- numbers:end -->` markers is generated; everything else in the two
files is prose that cites A/B files of its own."""
import json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
b = json.load(open(os.path.join(R, "profiles", f"{tag}_bench.json")))
t = json.load(open(os.path.join(R, "profiles", f"{tag}_traffic.json")))["configs"]
def sp(x, nd=0): return f"{x:,.{nd}f}".replace(",", " ")
def kern(cfg, pat): return [v for k, v in t[cfg]["per_kernel"].items() if pat in k]
rf, ex = b["roofline"], b["extra"]
sus, c5, c2, c4 = ex["c3_sustained"], ex["c5_4096x1080p"], ex["c2_idct_dequant_only"], ex["c4_encode_4096"]
c3k, c2k, c4k = kern("c3", "k_quad420")[0], kern("c2", "k_idct_plane")[0], kern("c4", "k_encode_fused")[0]
c5ks = kern("c5", "k_quad420")
c5x = t["c5"]["hbm_bytes_per_step"] / t["c5"]["algorithmic_bytes_per_step"]
c3x = (rf["traffic"] or t["c3"]["hbm_bytes_per_step"]) / rf["algorithmic_bytes_per_step"]
drv = [float(l.split()[2]) for l in open(os.path.join(R, "profiles", f"{tag}_bench_driver_args.txt")) if len(l.split()) == 3]
r8 = ex.get("c3_ring8", {})
design = f"""Round-6 result (`profiles/{tag}_bench.json`; `profiles/{tag}_c3_*` are the rocprofv3 files of the same tree, same box):
**{sp(b['value'])} Mpixels/s**, {b['ms_per_step']:.4f} ms/step, `roofline.achieved` = {sp(rf['achieved'])} GB/s = **{100 * rf['frac']:.1f} % of 8 TB/s**; the 2 000
further steps of `extra.c3_sustained` run at {sus['ms_per_step_median']:.4f} ms (median of ten windows; {sus['ms_per_step_min']:.4f}-{sus['ms_per_step_max']:.4f}) =
{100 * sus['frac_hbm_median']:.1f} %.  One launch of `k_quad420<1, 32, true, false>` per step ({c3k['mean_us']:.1f} µs mean under the profiler), {(rf['traffic'] or t['c3']['hbm_bytes_per_step']) / 1e6:.1f} MB of
measured traffic = {c3x:.2f} × algorithmic; **binding roofline: VALU issue, `roofline.valu_frac` = {rf.get('valu_frac', 0):.2f}** (§6.1; {rf.get('valu', {}).get('valu_frac_at_occupancy', 0):.2f} of what three waves
per SIMD can issue).  With the driver's own arguments (`--steps 20 --warmup 5`) the same box gives {100 * min(drv):.1f}-{100 * max(drv):.1f} % (`profiles/{tag}_bench_driver_args.txt`, §6.3);
over a ring of eight image sets instead of four (the footprint of rounds 1-4) {r8.get('ms_per_step', 0):.4f} ms = {100 * r8.get('frac_hbm', 0):.1f} % (`extra.c3_ring8`).  The decode kernel is round 5's:
the round's work went into the encode, the 4:4:0 walk and the generic kernels (§10), and the boxes of the pool differ by more than a round's gain (§6.3).

| config | what | time | algorithmic rate | of 8 TB/s | evidence |
|---|---|---|---|---|---|
| C3 | one 8192² 4:2:0 image → RGB8, `k_quad420<1, 32, true, false>` (static walk), one launch; VALU-bound ({rf.get('valu_frac', 0):.2f}) | {c3k['mean_us']:.1f} µs (rocprofv3 mean of {c3k['launches']}), {b['ms_per_step'] * 1e3:.1f} µs wall per step | {sp(rf['achieved'])} GB/s | **{100 * rf['frac']:.1f} %** | `profiles/{tag}_c3_kernel_stats.csv`, `{tag}_c3_pmc_*.csv`, `{tag}_bench.json` |
| C5 | 4096 × 1080p on one GPU (the N = 1 point of the `--gpus N` job): `k_quad420<1, 32, true, true>` over seven columns of 32 × 2 strips + `k_quad420<1, 16, true, true>` over the one column of 16 × 4 strips (ticket walk) ({' + '.join(f"{k['mean_us'] / 1e3:.2f}" for k in sorted(c5ks, key=lambda k: -k['mean_us']))} ms under the profiler), {c5x:.2f} × the algorithmic bytes; every rank checks one image of its shard against the batch and the oracle | {c5['ms']:.2f} ms | {sp(c5['GB_per_s'])} GB/s | **{100 * c5['GB_per_s'] / 8000:.1f} %** | `profiles/{tag}_c5_*`, `{tag}_traffic.json` |
| C2 | IDCT + dequant only, 2²² blocks, `k_idct_plane` | {c2k['mean_us']:.1f} µs | {sp(1073741824 / c2k['mean_us'] / 1e3)} GB/s | {100 * 1073741824 / c2k['mean_us'] / 8e6:.1f} % (= the streaming ceiling of §6.1) | `profiles/{tag}_c2_*` |
| C4 | encode 4096² RGB8 → 4:2:0 coefficients, `k_encode_fused<…, 8>`; VALU-bound ({c4.get('valu_frac', 0):.2f}) | {c4k['mean_us']:.1f} µs | {sp(100663296 / c4k['mean_us'] / 1e3)} GB/s | **{100 * 100663296 / c4k['mean_us'] / 8e6:.1f} %** | `profiles/{tag}_c4_*`, `{tag}_pmc_encode.txt`; coefficients equal the oracle's |
"""
readme = f"""Round 6 on one MI355X (`profiles/{tag}_bench.json`, `DESIGN.md` §6): 8192×8192 ycc8 4:2:0 → RGB8 in {b['ms_per_step']:.4f} ms per step
({b['value'] / 1e3:.0f} Gpixel/s, {rf['achieved'] / 1e3:.2f} TB/s algorithmic = **{100 * rf['frac']:.1f} %** of the 8 TB/s HBM peak; {100 * sus['frac_hbm_median']:.1f} % over 2 000 sustained steps,
{100 * ex['c3_two_streams']['frac_hbm_median']:.1f} % when two calls are in flight; round 5: 66.9 %, round 4: 62.5 %, round 3: 62.3 %, round 2: 55.9 %, round 1: 52.7 %) — one launch of `k_quad420`, bound by VALU issue
(`roofline.valu_frac` {rf.get('valu_frac', 0):.2f}: every float operation of the reference is its own instruction), that moves {c3x:.2f} × the
algorithmic bytes at {100 * rf.get('achieved_over_d2d_memcpy', 0):.0f} % of the rate the vendor's device-to-device memcpy reaches on the same box; 4096 × 1080p on one GPU in
{c5['ms']:.2f} ms (**{100 * c5['GB_per_s'] / 8000:.1f} %**, round 5: 72.3 %; {c5x:.2f} × the algorithmic bytes); IDCT + dequant alone {100 * c2['frac_hbm']:.1f} %; grey decode 71-77 %; encode 4096² in
{c4['ms'] * 1e3:.1f} µs ({100 * c4['frac_hbm']:.1f} %; round 5: 23.3 µs).
"""
for name, text in (("DESIGN.md", design), ("README.md", readme)):
    p = os.path.join(R, name)
    s = open(p).read()
    m = re.search(r"<!-- numbers:begin -->\n.*?<!-- numbers:end -->\n", s, re.S)
    assert m, f"{name}: markers missing"
    s = s[:m.start()] + "<!-- numbers:begin -->\n" + text + "<!-- numbers:end -->\n" + s[m.end():]
    open(p, "w").write(s)
    print(name, "updated")
