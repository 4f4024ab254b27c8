#!/usr/bin/env python3
"""Per-phase wall cycles of k_quad420's waves (one 8192 x 8192 4:2:0 image) (needs a -DJA_PHASE_PROFILE build):
    tools/build_exp.sh prof -DJA_PHASE_PROFILE
    JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_prof.so python tools/phase_profile.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth

ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
W, H, N = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8192, 8192, 1)   # [warm] [W H N]
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
planes = [synth.natural_planes_torch(units, N, dev, 3 + r) for r in range(2)]
out = torch.empty((2, N * W * H * 3), dtype=torch.uint8, device=dev)
strides = _lib.size_array([64 * a * b for a, b in units])
def step(r):
    st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), N, _lib.ptr_array([p.data_ptr() for p in planes[r]]), strides,
                                   d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out[r].data_ptr(), W * H * 3)
    assert st == 0
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 4   # steps before the measured one (200: the power cap has bitten)
for i in range(warm): step(i & 1)
torch.cuda.synchronize()
ctx.timer_begin(); step(0); ms = ctx.timer_end()
fn = lib.jpeg_amd_debug_phase_cycles; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((4096, 16), np.uint64)
assert fn(buf.ctypes.data, buf.size) == 0
buf = buf[buf[:, 14] > 0]   # waves that ran (3 072 at three waves per SIMD, 4 096 at four)
print(f"{len(buf)} waves")
names = ["wait chroma coef DMA", "chroma block read + luma DMA issue", "chroma IDCT + pack", "wait for the others' reads (WAR)",
         "tile write + arrive", "patch row requests", "luma IDCT", "hrow prologue + geometry",
         "pixel rows 1-5", "wait for the stack's tile (RAW)", "halo repair, rows 6, 0, 7, last stores", "luma DMA wait (vmcnt 0)",
         "luma block read + lgkmcnt(0)", "next chroma DMA issue"]
def _stacks(cols, uy, by): return cols * ((-(-uy // by) + 3) // 4)
def _cut(ux, uy, n):   # quad_cut in kernels_quad.hip
    wide, narrow = _stacks(-(-ux // 32), uy, 2), _stacks(-(-ux // 16), uy, 4)
    whole, rest = ux // 32, ux % 32
    if whole and 0 < rest <= 16 and _stacks(whole, uy, 2) + _stacks(1, uy, 4) < min(wide, narrow) and n * min(wide, narrow) >= 16 * 768:
        return _stacks(whole, uy, 2) + _stacks(1, uy, 4)
    return min(wide, narrow)
nstrips = 4 * N * _cut(units[0][0], units[0][1], N)   # strip slots of the call (the LAST launch's waves are what the profile arrays hold)
tot = buf[:, :14].sum(axis=1).astype(np.float64)
print(f"step {ms*1e3:.1f} us; per-wave total cycles mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(names):
    c = buf[:, i].astype(np.float64)
    print(f"  {n:40s} {c.mean():10.0f} cycles/wave  {100*c.mean()/tot.mean():5.1f} %   per strip {c.mean()/(nstrips/len(buf)):8.0f}")
life, ticks = buf[:, 14].astype(np.float64), buf[:, 15].astype(np.float64)
print(f"wave life: mean {life.mean():.0f} max {life.max():.0f} shader cycles = mean {ticks.mean() / 100:.1f} max {ticks.max() / 100:.1f} us of the 100 MHz counter"
      f" -> effective shader clock {100.0 * (life / ticks).mean():.0f} MHz while k_luma_fused runs")
# per-wave placement and timing (start / end on the chip-wide 100 MHz counter, HW_ID, XCC_ID)
fi = lib.jpeg_amd_debug_wave_info; fi.restype = C.c_int; fi.argtypes = [C.c_void_p, C.c_size_t]
wi = np.zeros((4096, 4), np.uint64)
if fi(wi.ctypes.data, wi.size) == 0:
    wi = wi[:len(buf)].astype(np.int64)
    t0 = wi[:, 0].min()
    start, end = (wi[:, 0] - t0) / 100.0, (wi[:, 1] - t0) / 100.0
    print(f"wave start: min {start.min():.1f} mean {start.mean():.1f} max {start.max():.1f} us; end: min {end.min():.1f} mean {end.mean():.1f} max {end.max():.1f} us")
    print("end-time percentiles (us):", " ".join(f"{p}%={np.percentile(end, p):.1f}" for p in (1, 10, 25, 50, 75, 90, 99, 100)))
    xcc = wi[:, 3]
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"  XCC {x}: {m.sum():5d} waves  start mean {start[m].mean():6.1f}  end mean {end[m].mean():6.1f} max {end[m].max():6.1f}  life mean {(end[m]-start[m]).mean():6.1f} min {(end[m]-start[m]).min():6.1f} max {(end[m]-start[m]).max():6.1f} us  clock {100.0*(life[m]/ticks[m]).mean():.0f} MHz")
    hw = wi[:, 2]
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    print("HW_ID samples:", [hex(int(h)) for h in hw[:8]])
    key = xcc * 1000 + se * 100 + sh * 16 + cu
    ends_by_cu = {}
    for k, e in zip(key, end): ends_by_cu.setdefault(int(k), []).append(e)
    cu_end = np.array([max(v) for v in ends_by_cu.values()]); cu_n = np.array([len(v) for v in ends_by_cu.values()])
    print(f"{len(cu_end)} distinct (xcc, se, sh, cu); waves per CU min {cu_n.min()} max {cu_n.max()}; last wave of a CU ends: min {cu_end.min():.1f} mean {cu_end.mean():.1f} max {cu_end.max():.1f} us")
    # workgroups that walk one stack more than the others (slot = 4 * blockIdx + wave): do they end later?
    nwg = len(buf) // 4; nst = nstrips // 4; n_long = nst % nwg
    gen = (np.arange(len(buf)) // 4) * 3 // nwg   # the dispatcher places workgroups in blockIdx order: three generations of one per CU
    for g in range(3):
        m = gen == g
        print(f"  generation {g}: start mean {start[m].mean():6.1f} end mean {end[m].mean():8.1f} min {end[m].min():8.1f} max {end[m].max():8.1f} us")
    per_cu_g = {}
    for k, g in zip(key, gen): per_cu_g.setdefault(int(k), set()).add(int(g))
    print(f"  CUs that hold all three generations: {sum(len(v) == 3 for v in per_cu_g.values())} of {len(per_cu_g)}")
    if n_long:
        wg = np.arange(len(buf)) // 4; lng = wg < n_long
        print(f"{n_long} workgroups walk {nst // nwg + 1} stacks, {nwg - n_long} walk {nst // nwg}")
        for nm, m in (("long", lng), ("short", ~lng)):
            print(f"  {nm:5s}: end mean {end[m].mean():6.1f} min {end[m].min():6.1f} max {end[m].max():6.1f} us; percentiles " + " ".join(f"{p}%={np.percentile(end[m], p):.1f}" for p in (10, 50, 90)))
        per_cu = {}
        for k, l in zip(key, lng): per_cu.setdefault(int(k), []).append(bool(l))
        cnt = np.array([sum(v) // 4 for v in per_cu.values()])
        print("  long workgroups per CU: " + " ".join(f"{c}:{(cnt == c).sum()}" for c in sorted(set(cnt))))
        # time the last long workgroup of a CU runs after the CU's last short wave ended
        gap = []
        for k in per_cu:
            m = key == k
            if (m & lng).any() and (m & ~lng).any(): gap.append(end[m & lng].max() - end[m & ~lng].max())
        gap = np.array(gap); print(f"  per CU: last long end - last short end: mean {gap.mean():.1f} min {gap.min():.1f} max {gap.max():.1f} us")
