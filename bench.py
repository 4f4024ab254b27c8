#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spectral pipeline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|c3|c5] [--no-extras]

Metric (BASELINE.json): Mpixels/s of device-resident fused decode
(dequant + IDCT + 4:2:0 upsample + YCbCr->RGB), and its fraction of the HBM roofline.
A step = one pass of jpeg_amd_decode_batch over one batch of synthetic coefficient planes
already resident in HBM (host Huffman decoding and PCIe transfers are out of scope and
excluded; see DESIGN.md "Measurement").

Workload c3 (BASELINE.json configs[2], the configuration the metric is quoted on; `value` at EVERY N): one
8192x8192 ycc8 4:2:0 image per GPU per step; the step rotates through a ring of distinct images so that the
256 MiB Infinity Cache cannot serve the input.  With --gpus N > 1 every rank decodes its own image: weak scaling,
the same per-GPU work at every N, so a 1/2/4/8 curve built from `value` compares like with like.
The C5 job (configs[4]): ONE job of 4096 independent 1920x1080 images, sharded over the ranks with
jpeg_amd.dist.shard (contiguous chunks: 512 per GPU at N = 8, all 4096 = 51 GB of coefficients + pixels on the one
GPU at N = 1) -- strong scaling -- is measured collectively (barrier, max over ranks) at EVERY N right after the
headline and reported as `extra.c5_4096x1080p`; a strong-scaling curve is built from ITS `Mpixels_per_s`.
`--workload c5` makes that job the headline instead (`value`, "scaling": "strong").
Every rank proves what it timed: one image of its shard is decoded again on its own, compared with the batch result on
the device and with the CPU oracle on the host (`per_rank[i].parity_vs_oracle`, `single_equals_batch`), for the
headline and for the C5 job.
There is no data-path collective in either workload: every rank decodes its own independent images;
the only collective is an RCCL broadcast of the quantisation tables from rank 0 before the timed
region.

For N > 1 launch with:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6290.0  # measured float4 copy ceiling, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto", choices=["auto", "c3", "c5"],
                    help="auto = c3 at every N (weak scaling, the headline); c5: the sharded 4096 x 1080p job as the headline")
    ap.add_argument("--dist", action="store_true",
                    help="initialise torch.distributed also at N = 1 (RCCL world of one: the table broadcast and the max over ranks then run as real collectives)")
    ap.add_argument("--no-c5-job", action="store_true", help="skip the collective C5 job leg (extra.c5_<n>x1080p)")
    ap.add_argument("--c5-steps", type=int, default=5)
    ap.add_argument("--no-parity", action="store_true", help="skip the per-rank comparison with the CPU oracle")
    ap.add_argument("--c5-images", type=int, default=4096, help="images of the c5 job (all ranks together)")
    ap.add_argument("--ring", type=int, default=0, help="distinct image sets to rotate through")
    ap.add_argument("--no-extras", action="store_true", help="skip the C2/C4/C5 side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--traffic", default="live", choices=["live", "file", "none"],
                    help="roofline.traffic: measured now under rocprofv3 (c3, N = 1), profiles/traffic_latest.json, or null")
    ap.add_argument("--valu", default="live", choices=["live", "none"],
                    help="roofline.valu / extra.c4_encode_4096.valu: the VALU issue roofline of k_quad420 and k_encode_fused, measured now "
                         "under rocprofv3 (N = 1), or omitted")
    return ap.parse_args()


class DecodeWorkload:
    """Fused decode of `n_images` identically laid out ycc8 4:2:0 images per step."""

    def __init__(self, J, ctx, width, height, n_images, ring, quanta, seed):
        import torch
        from jpeg_amd import synth, _lib
        self.J, self.ctx, self._lib = J, ctx, _lib
        self.size = (width, height)
        self.n_images, self.ring = n_images, ring
        self.layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1),
                                        3: J.Component((1, 1), 1)})
        self.units = self.layout.units(self.size)
        self.L = self.layout.c_layout(self.size, self.units, [0, 1, 1])
        dev = ctx.torch_device
        # ring * n_images images per plane, distribution N
        self.planes = synth.natural_planes_torch(self.units, ring * n_images, dev, seed)
        self.stride = [64 * ux * uy for ux, uy in self.units]
        self.pixel_stride = width * height * 3
        self.out = torch.empty((ring * n_images, self.pixel_stride), dtype=torch.uint8, device=dev)
        self.d_quanta = quanta            # device tensor int16 view of uint16 [2][64]
        self.blocks = sum(ux * uy for ux, uy in self.units)
        self.pixels_per_image = width * height
        self.pixels = width * height * n_images
        # algorithmic bytes per step: coefficients in + RGB8 out
        self.bytes = (128 * self.blocks + 3 * width * height) * n_images
        self._args = []
        for r in range(ring):
            ptrs = _lib.ptr_array([p[r * n_images].data_ptr() for p in self.planes])
            self._args.append((ptrs, self.out[r * n_images].data_ptr()))
        self._strides = _lib.size_array(self.stride)
        self._fn = _lib.lib().jpeg_amd_decode_batch
        self._step = 0

    def step(self):
        ptrs, out = self._args[self._step % self.ring]
        self._step += 1
        st = self._fn(self.ctx.handle, C.byref(self.L), self.n_images, ptrs, self._strides,
                      self.d_quanta.data_ptr(), 0, 2, 0, self._lib.COLOR_RGB8, out,
                      self.pixel_stride)
        if st != 0:
            raise self._lib.JpegAmdError(st, "jpeg_amd_decode_batch", 0)

    def timer_begin(self):
        self.ctx.timer_begin()

    def timer_end(self):
        return self.ctx.timer_end()

    def device_name(self):
        import torch
        return torch.cuda.get_device_name(self.ctx.torch_device)

    def verify(self, quanta_np, image=0, threads=8):
        """What this rank just timed, proven on one image of its shard (image `image` of ring slot 0, which every step count
        >= 1 has decoded): decoded again ON ITS OWN (a batch of one) and compared with the batch result on the device, then
        compared with the CPU oracle on the host.  Every pixel is a function of its own image's coefficients only
        (decode.swift:4114-4117), so one image per rank is a witness of the whole shard's code path."""
        import torch
        res = {"image_checked": int(image), "single_equals_batch": None}
        if self.n_images > 1:
            alone = torch.empty(self.pixel_stride, dtype=torch.uint8, device=self.out.device)
            ptrs = self._lib.ptr_array([p[image].data_ptr() for p in self.planes])
            st = self._fn(self.ctx.handle, C.byref(self.L), 1, ptrs, self._strides, self.d_quanta.data_ptr(), 0, 2, 0,
                          self._lib.COLOR_RGB8, alone.data_ptr(), self.pixel_stride)
            if st != 0:
                raise self._lib.JpegAmdError(st, "jpeg_amd_decode_batch", 0)
            res["single_equals_batch"] = bool(torch.equal(alone, self.out[image]))
        planes, pixels = self.host_case(image)
        res["parity_vs_oracle"] = oracle_check(planes, pixels, self.size, quanta_np, threads)
        return res

    def host_case(self, image=0):
        """Coefficient planes and the pixels the device produced for one image of the LAST step:
        what the cpu_baseline leg decodes again on the host and compares (not timed)."""
        planes = [p[image].cpu().numpy() for p in self.planes]     # ring slot 0 (every slot has been decoded)
        return planes, self.out[image].cpu().numpy().reshape(-1, 3)


def time_region(wl, fn, steps, sync, barrier, errors=None):
    """Barrier + synchronize on both sides; returns (wall seconds, device-event ms: HIP events on the stream the
    kernels are launched on, via the workload's timer).  With `errors` (a list) an exception of the local work is
    recorded there instead of being raised, and the rank STILL meets both barriers: the collective sequence of a
    measurement is the same on every rank whatever happens to one of them."""
    barrier()
    sync()
    t0 = time.perf_counter()
    gpu_ms = float("nan")
    try:
        if wl is not None:
            wl.timer_begin()
        for _ in range(steps):
            fn()
        if wl is not None:
            gpu_ms = wl.timer_end()
        sync()
    except Exception as e:
        if errors is None:
            raise
        errors.append(repr(e)[:200])
    barrier()
    return time.perf_counter() - t0, gpu_ms


def oracle_check(planes, pixels, size, quanta_np, threads):
    """Checker half of the cpu_baseline leg (oracle/ is test infrastructure and is only ever the CHECKER here, never the
    thing timed as the product): decode `planes` (ycc8 4:2:0) on the host and compare with the device's RGB8 `pixels`."""
    from oracle import oracle as O
    _, rect = O.decode(planes, [quanta_np[0], quanta_np[1], quanta_np[1]], [(2, 2), (1, 1), (1, 1)], tuple(size), threads=threads)
    return bool((O.unpack_rgb8(rect, 3, threads=threads) == pixels.reshape(-1, 3)).all())


def cpu_baseline(J, quanta_np, seconds, device_case=None, encode_case=None):
    """The ONLY place of this file that touches oracle/ (test infrastructure): times the oracle
    -- the C restatement of the reference's CPU algorithm -- on this host on a bounded sample of
    the same workload shape (ycc8 4:2:0 fused decode to RGB8), and, while it has CPU results in
    hand, compares them with what the device produced (`device_case`: planes + pixels of one
    timed image; `encode_case`: pixels + coefficient planes of the config-4 frame)."""
    import numpy as np
    from jpeg_amd import synth
    from oracle import oracle as O

    def make(w, h):
        layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
        units = layout.units((w, h))
        planes = [synth.blocks_natural(ux * uy, 11 + i).reshape(uy, ux, 64) for i, (ux, uy) in enumerate(units)]
        return planes

    def run(planes, size, threads):
        t0 = time.perf_counter()
        _, rect = O.decode(planes, [quanta_np[0], quanta_np[1], quanta_np[1]],
                           [(2, 2), (1, 1), (1, 1)], size, threads=threads)
        O.unpack_rgb8(rect, 3, threads=threads)
        return time.perf_counter() - t0

    probe = (512, 512)
    p = make(*probe)
    run(p, probe, 1)
    t = run(p, probe, 1)
    rate1 = probe[0] * probe[1] / t                          # px/s, one thread
    side = int(min(8192, max(512, (rate1 * seconds) ** 0.5)) // 16 * 16)
    size = (side, side)
    planes = make(*size)
    t1 = run(planes, size, 1)
    reps = max(1, min(32, int(round(seconds / t1))))         # about `seconds` of single-core work
    if reps > 1:
        t1 = sum(run(planes, size, 1) for _ in range(reps)) / reps
    ncores = host_cpus()
    threads = min(ncores, 64)
    run(planes, size, threads)
    tn = min(run(planes, size, threads) for _ in range(3))
    mpx = size[0] * size[1] / 1e6
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        model = "unknown"
    parity = {}
    nthr = min(64, host_cpus())
    if device_case is not None:
        planes_d, pixels_d, size_d = device_case
        _, rect = O.decode(planes_d, [quanta_np[0], quanta_np[1], quanta_np[1]], [(2, 2), (1, 1), (1, 1)], size_d, threads=nthr)
        parity["decode_equals_cpu"] = bool((O.unpack_rgb8(rect, 3, threads=nthr) == pixels_d).all())
    if encode_case is not None:
        rgb_e, coef_e, size_e = encode_case
        want = O.encode(rgb_e, size_e, [(2, 2), (1, 1), (1, 1)], [quanta_np[0], quanta_np[1], quanta_np[1]], threads=nthr)
        parity["encode_equals_cpu"] = bool(all((c.reshape(w.shape) == w).all() for c, w in zip(coef_e, want)))
    return {
        "parity": parity,
        "value": round(mpx / t1, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"{reps} x one {size[0]}x{size[1]} ycc8 4:2:0 image, distribution N, fused decode to RGB8 "
                  f"(oracle/jpeg_oracle.c, gcc -O2 -ffp-contract=off), {t1:.2f} s each, {reps * t1:.1f} s in all",
        "all_cores": {"value": round(mpx / tn, 3), "cores": threads, "seconds": round(tn, 3)},
        "host": {"model": model, "nproc": ncores},
        "note": "C restatement of tayloraswift/jpeg's CPU algorithm (the Swift toolchain is absent); "
                "a reported baseline, not the optimisation target",
    }


DECODE_SOURCES = ("kernels_quad.hip", "kernels_fused.hip", "fused_common.hpp", "dct.hpp", "upsample.hpp", "kernels.hpp", "capi.hip")   # (tools/make_traffic.py hashes the same list)


def kernel_source_sha16():
    """Digest of the device sources of the decode path: a `roofline.traffic` taken from profiles/traffic_latest.json (the
    fallback when the live measurement below is unavailable) is only reported while it belongs to the kernels being timed."""
    import hashlib
    h = hashlib.sha256()
    for f in DECODE_SOURCES:
        h.update(open(os.path.join(ROOT, "jpeg_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def under_profiler() -> bool:
    """True when this process itself runs under rocprofv3 (tools/profile_round.sh): a nested profiler run would be an exec of a
    launcher from a process whose preloaded tool library has already initialised the GPU, which the GPU boxes refuse."""
    return any(k in os.environ for k in ("ROCPROFILER_LIBRARY_CTOR", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or \
        "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def measure_traffic_live(timeout=240):
    """HBM bytes per launch of the C3 step's kernel, measured NOW on this GPU: two child runs of tools/run_c3.py (the same
    jpeg_amd_decode_batch call on the same 8192 x 8192 workload) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` -- separate passes, as MI355X_MICROARCH.md's HBM section prescribes; both counters are in KiB and
    FETCH_SIZE tallies the 128-byte requests of 16-B/lane streaming reads at 64 B on gfx950, so it is doubled.
    -> (bytes per launch, description) or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if under_profiler():
        return None, "this run is itself being profiled"
    if os.environ.get("JPEG_AMD_LIBRARY"):
        return None, "JPEG_AMD_LIBRARY is set: the timed library is not the product build"
    out = {}
    tmp = tempfile.mkdtemp(prefix="jpeg_amd_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "run_c3.py"), "24"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]!r}"
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if row.get("Counter_Name") == counter and "k_quad420" in row.get("Kernel_Name", "")]
            if not vals:
                return None, f"no k_quad420 dispatch in the {counter} pass"
            out[counter] = (sum(vals) / len(vals), len(vals))
    except Exception as e:   # never let the measurement break the headline line
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    nbytes = int(round((2.0 * out["FETCH_SIZE"][0] + out["WRITE_SIZE"][0]) * 1024))
    return nbytes, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/run_c3.py, "
                    f"mean of {out['FETCH_SIZE'][1]} / {out['WRITE_SIZE'][1]} k_quad420 launches; bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB")


def measure_valu_live(which):
    """The binding roofline (tools/valu_roofline.py): SQ instruction-class counters of the C3 / C4 kernel, measured NOW by two
    rocprofv3 child runs of the same call on the same workload, turned into VALU issue cycles needed per SIMD / cycles elapsed."""
    try:
        from tools import valu_roofline as vr
        if under_profiler():
            return {"error": "this run is itself being profiled"}
        if os.environ.get("JPEG_AMD_LIBRARY"):
            return {"error": "JPEG_AMD_LIBRARY is set: the timed library is not the product build"}
        lib = os.path.join(ROOT, "jpeg_amd", "libjpeg_amd.so")
        if which == "c3":
            return vr.measure([sys.executable, os.path.join(ROOT, "tools", "run_c3.py"), "24"], "k_quad420<1, 32, true, false>", lib,
                              "k_quad420ILi1ELi32ELb1ELb0", waves_per_simd=3)
        return vr.measure([sys.executable, os.path.join(ROOT, "tools", "bench_encode.py"), "--only", "4:2:0", "--reps", "24"],
                          "k_encode_fused<2, 2, true, true, true, 8, false>", lib, "k_encode_fusedILi2ELi2ELb1ELb1ELb1ELi8ELb0", waves_per_simd=4)
    except Exception as e:   # never let the measurement break the headline line
        return {"error": repr(e)[:300]}


def run(args, make_workload=None, backend="nccl", device_kind="cuda"):
    """One rank of the benchmark.  `make_workload(name, width, height, n_images, ring, quanta, seed)` builds the
    object whose step() is timed (default: DecodeWorkload on this rank's GPU); tests/ drives this same function
    under gloo on CPU with an oracle-backed workload.  Returns the result dict on rank 0, None elsewhere."""
    import numpy as np
    import torch
    import jpeg_amd as J
    from jpeg_amd import dist as jd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 with python -m torch.distributed.run "
                         "(see module docstring)")
    if device_kind == "cuda":
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")

    dist = None
    if world > 1 or getattr(args, "dist", False):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if device_kind == "cuda":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world and dist.get_rank() == rank

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        if device_kind == "cuda":
            torch.cuda.synchronize(dev)

    ctx = J.Context(local) if device_kind == "cuda" else None

    # quantisation tables: rank 0 owns them, RCCL broadcast over xGMI is the only collective
    q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
    d_quanta = jd.broadcast_quanta(q_np if rank == 0 else 2, 0, dev, dist)
    if dist is not None:
        assert (d_quanta.cpu().numpy().view(np.uint16) == q_np).all()

    if not args.no_parity:
        # the checker (oracle/: a C restatement built with gcc) is compiled by ONE rank; the others load the finished library
        if rank == 0:
            from oracle import oracle as _O
            _O.build()
        barrier()

    workload = args.workload if args.workload != "auto" else "c3"   # ONE workload under `value` at every N
    check_threads = max(1, min(64, host_cpus() // world))
    if make_workload is None:
        def make_workload(name, width, height, n_images, ring, quanta, seed):
            return DecodeWorkload(J, ctx, width, height, n_images, ring, quanta, seed)
    if workload == "c3":
        # Ring of distinct image sets, so that the 256 MiB Infinity Cache cannot serve the input: four sets = 1.6 GB of coefficients
        # and pixels between two uses of a set.  (Rounds 1-4 used eight: 3.2 GB of mappings exceed the GPU's TLB reach and the step
        # then pays ~3 us of page walks that belong to the benchmark's footprint, not to the decode of ONE image -- every set alone
        # decodes in 74-77 us, rotating over 4 sets 75.4, over 8 sets 78.0: profiles/r05_bench_ring.txt.  The sharded C5 job below,
        # 51 GB per step, carries the full cost of a large footprint.)
        ring = args.ring or 4
        wl = make_workload("c3", 8192, 8192, 1, ring, d_quanta, 20240807 + 1000 * rank)
        scaling, images_total = "weak", world
        name = ("C3: one 8192x8192 ycc8 4:2:0 image per GPU per step, fused Spectral->RGB8 decode "
                f"(ring of {ring} distinct images)")
        parallelism = f"independent images x{world} (weak scaling), no data-path collective"
    else:
        # ONE job of args.c5_images independent images, contiguous shards (jpeg_amd.dist.shard)
        lo, hi = jd.shard(args.c5_images, rank, world)
        ring = args.ring or (2 if (hi - lo) <= 1024 else 1)   # 12.4 MB per image and ring slot; the set is >> 256 MiB anyway
        wl = make_workload("c5", 1920, 1080, hi - lo, ring, d_quanta, 20240807 + lo)
        scaling, images_total = "strong", args.c5_images
        name = (f"C5: {args.c5_images} independent 1920x1080 ycc8 4:2:0 images per step, sharded over {world} GPU(s) "
                f"({hi - lo} on rank 0), fused Spectral->RGB8 decode (ring of {ring} distinct batches)")
        parallelism = f"images sharded contiguously over {world} rank(s) (strong scaling), no data-path collective"

    # set-up, not warm-up: the first call of a process loads the code object and sizes the persistent grid (once per
    # kernel instantiation), and the first use of a buffer pays for its page-table walks (a ring slot that is decoded for the
    # first time INSIDE a 20-step timed region costs ~30 us, 1.5 steps' worth: measured, profiles/r05_bench_ring.txt).  One call
    # per ring slot is made here, so that `--warmup 0` times the hot path and not the loader or the memory manager; the
    # W warm-up steps and the K timed steps that follow are what the contract asks for
    for _ in range(max(1, getattr(wl, "ring", 1))):
        wl.step()
    sync()
    wl._step = 0
    for _ in range(args.warmup):
        wl.step()
    wall, gpu_ms = time_region(wl, wl.step, args.steps, sync, barrier)
    wall_max = jd.max_over_ranks(wall, dev, dist)

    # per-rank record (device, images, times, and the proof of what was timed), gathered on every rank; rank 0 reports it
    mine = {"rank": rank, "device": wl.device_name(), "images_per_step": wl.n_images,
            "wall_ms_per_step": round(wall / args.steps * 1e3, 5), "gpu_ms_per_step": round(gpu_ms / args.steps, 5)}
    if not args.no_parity and wl.n_images > 0:
        mine.update(wl.verify(q_np, image=(wl.n_images - 1) // 2, threads=check_threads))
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        assert sum(r["images_per_step"] for r in per_rank) == images_total

    # sustained behaviour (the step runs at the package power limit): ten more windows of 200 steps, not part of `value`
    sustained = None
    if world == 1 and workload == "c3" and device_kind == "cuda" and not args.no_extras:
        win = []
        for _ in range(10):
            w_s, _ = time_region(wl, wl.step, 200, sync, barrier)
            win.append(w_s / 200 * 1e3)
        win.sort()
        sustained = {"steps": 2000, "windows": 10, "ms_per_step_min": round(win[0], 5), "ms_per_step_median": round((win[4] + win[5]) / 2, 5),
                     "ms_per_step_max": round(win[-1], 5),
                     "frac_hbm_median": round(wl.bytes / ((win[4] + win[5]) / 2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "frac_hbm_min_max": [round(wl.bytes / (win[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), round(wl.bytes / (win[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)],
                     "note": "2000 further steps right after the timed region, wall clock per window of 200"}

    # the same steps issued alternately on TWO contexts (two streams): what an application gets by double-buffering its
    # decodes -- every launch fills the chip, so the next one's workgroups start as this one's leave and the tail of one
    # overlaps the head of the other.  Not part of `value` (one stream, one launch after the other).
    two_streams = None
    if sustained is not None:
        try:
            ctx2 = J.Context(0, own_stream=True)
            fn, L, strides = wl._fn, wl.L, wl._strides
            def step2(i):
                ptrs, out = wl._args[i % wl.ring]
                st = fn((ctx if i & 1 else ctx2).handle, C.byref(L), wl.n_images, ptrs, strides, wl.d_quanta.data_ptr(), 0, 2, 0,
                        wl._lib.COLOR_RGB8, out, wl.pixel_stride)
                if st != 0:
                    raise wl._lib.JpegAmdError(st, "jpeg_amd_decode_batch", 0)
            win = []
            for _ in range(6):
                sync(); t0 = time.perf_counter()
                for i in range(200):
                    step2(i)
                sync(); win.append((time.perf_counter() - t0) / 200 * 1e3)
            win.sort()
            two_streams = {"steps": 1200, "windows": 6, "ms_per_step_median": round((win[2] + win[3]) / 2, 5), "ms_per_step_min": round(win[0], 5),
                           "frac_hbm_median": round(wl.bytes / ((win[2] + win[3]) / 2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "note": "the steps of c3_sustained issued alternately on two contexts (two streams), wall clock per window of 200"}
            ctx2.close()
        except Exception as e:   # a side measurement must not take the bench line with it
            two_streams = {"error": repr(e)[:200]}

    # the same 20-step window over a ring of EIGHT image sets (rounds 1-4 rotated over 8: 3.2 GB of mappings, beyond the GPU's TLB
    # reach; the default ring is 4 since round 5 -- ADVICE r05: report the old footprint beside the new one so that rounds stay
    # comparable).  Not part of `value`.
    ring8 = None
    if sustained is not None and wl.ring != 8:
        try:
            wl8 = make_workload("c3", 8192, 8192, 1, 8, d_quanta, 20240807 + 1000 * rank)
            for _ in range(8 + args.warmup):
                wl8.step()
            sync()
            w_s, _ = time_region(wl8, wl8.step, args.steps, sync, barrier)
            ring8 = {"ring": 8, "steps": args.steps, "ms_per_step": round(w_s / args.steps * 1e3, 5),
                     "frac_hbm": round(wl8.bytes / (w_s / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "the timed region repeated over 8 distinct image sets (the footprint of rounds 1-4's headline)"}
            del wl8
        except Exception as e:   # a side measurement must not take the bench line with it
            ring8 = {"error": repr(e)[:200]}

    # ---- the sharded C5 job (BASELINE.json configs[4]) at EVERY N, measured collectively: the strong-scaling curve is built
    #      from this record's Mpixels_per_s; `value` above stays on one workload ----
    c5_job = None
    if workload != "c5" and not args.no_c5_job:
        c5_job = run_c5_job(args, make_workload, d_quanta, q_np, rank, world, dev, dist, sync, barrier, check_threads)

    result = None
    if rank == 0:
        pixels_per_step = wl.pixels_per_image * images_total
        value = pixels_per_step * args.steps / wall_max / 1e6
        gpu_s_per_step = gpu_ms / 1e3 / args.steps
        # `achieved` / `frac` follow from the same clock as `value` (wall time between the barriers, launch gaps included);
        # the HIP-event time of rank 0's stream is the named secondary
        achieved = wl.bytes / (wall_max / args.steps) / 1e9
        achieved_events = wl.bytes / gpu_s_per_step / 1e9
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        mode = getattr(args, "traffic", "live")
        if mode == "live" and workload == "c3" and world == 1 and device_kind == "cuda":
            traffic, traffic_note = measure_traffic_live()
        if traffic is None and mode != "none" and os.path.exists(tpath) and device_kind == "cuda":
            live_note = traffic_note
            try:
                tj = json.load(open(tpath))
                per_step = tj.get("hbm_bytes_per_step") if workload == "c3" else None
                if workload == "c5" and "c5" in tj:     # measured on the whole 4096-image job: this rank's share of it
                    per_step = int(round(tj["c5"]["hbm_bytes_per_step"] * wl.n_images / tj["c5"]["images"]))
                if per_step is None:
                    traffic_note = f"profiles/traffic_latest.json has no record for workload {workload}"
                elif tj.get("kernel_source_sha16") != kernel_source_sha16():
                    traffic_note = ("stale: profiles/traffic_latest.json was measured at commit "
                                    f"{tj.get('commit')} with other kernel sources")
                else:
                    traffic = per_step
                    traffic_note = f"rocprofv3 PMC passes of this workload at commit {tj.get('commit')} (tools/profile_round.sh)"
            except Exception as e:
                traffic_note = repr(e)
            if live_note:
                traffic_note = f"{traffic_note}; live measurement unavailable: {live_note}"
        result = {
            "metric": "Mpixels/s decode (IDCT+dequant+upsample+YCbCr->RGB); % HBM roofline",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall_max / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "image": list(wl.size), "images_per_step_all_ranks": images_total,
                       "sampling": "4:2:0", "output": "RGB8", "coefficients": "distribution N (SURVEY 8d), seeded",
                       "quanta": "CompressionLevel luminance/chrominance(1.0)", "parallelism": parallelism},
            "per_rank": per_rank,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_note,
                         "algorithmic_bytes_per_step": wl.bytes,
                         "clock": "wall time of the timed region / steps (the clock of `value`)",
                         "gpu_ms_per_step_hip_events": round(gpu_s_per_step * 1e3, 5),
                         "achieved_hip_events": round(achieved_events, 1),
                         "frac_hip_events": round(achieved_events / HBM_PEAK_GBS, 4),
                         "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4),
                         "kernels": "all kernels of one fused decode step (rank 0's shard)"},
        }

    if rank == 0:
        result["scaling_curve"] = ("`value` is workload " + workload + " at every N (" + scaling + " scaling); the sharded C5 job is "
                                   "extra.c5_%dx1080p.Mpixels_per_s at every N (strong scaling)" % args.c5_images)
        checks = [r.get("parity_vs_oracle") for r in per_rank]
        result["parity_vs_oracle"] = None if any(c is None for c in checks) else bool(all(checks))
        if c5_job is not None:
            result.setdefault("extra", {})["c5_%dx1080p" % args.c5_images] = c5_job
    # ---- not timed: side measurements, CPU baseline (+ parity of what was just measured) ----
    if rank == 0 and device_kind == "cuda":
        encode_case = None
        if world == 1 and not args.no_extras:
            result.setdefault("extra", {}).update(extras(J, ctx, d_quanta, q_np, sync, args, workload))
            if sustained:
                result["extra"]["c3_sustained"] = sustained
            if two_streams:
                result["extra"]["c3_two_streams"] = two_streams
            if ring8:
                result["extra"]["c3_ring8"] = ring8
            encode_case = result["extra"].pop("_c4_host_case", None)
            # what the vendor's device-to-device memcpy moves on THIS box (read + written bytes per
            # second), measured just now: the practical ceiling of a 1 : 1 read / write stream
            d2d = result["extra"].pop("_d2d_copy_GBps", None)
            if d2d:
                rf = result["roofline"]
                rf["d2d_memcpy_live_GBps"] = d2d
                rf["achieved_over_d2d_memcpy"] = round(rf["achieved"] / d2d, 4)
                if rf["traffic"]:
                    rf["traffic_rate_over_d2d_memcpy"] = round(
                        rf["traffic"] / (result["ms_per_step"] * 1e-3) / 1e9 / d2d, 4)
        if world == 1 and not args.no_cpu:
            planes_h, pixels_h = wl.host_case(0)
            cb = cpu_baseline(J, q_np, args.cpu_seconds, (planes_h, pixels_h, wl.size), encode_case)
            parity = cb.pop("parity")
            result["cpu_baseline"] = cb
            result["gpu_over_cpu_1core"] = round(result["value"] / cb["value"], 1)
            if result.get("parity_vs_oracle") is None:
                result["parity_vs_oracle"] = parity.get("decode_equals_cpu")
            else:   # a second image (ring slot 0, image 0) checked by the baseline leg itself
                result["parity_vs_oracle"] = bool(result["parity_vs_oracle"] and parity.get("decode_equals_cpu"))
            if "encode_equals_cpu" in parity and "extra" in result:
                result["extra"]["c4_encode_4096"]["coefficients_equal_oracle"] = parity["encode_equals_cpu"]
        # ---- the BINDING roofline: both hot kernels are held by VALU issue (every float operation of the reference is its own
        #      instruction: no FMA contraction), so the HBM fraction alone does not say how close they are to what the chip can do
        if world == 1 and getattr(args, "valu", "live") == "live" and workload == "c3":
            rf = result["roofline"]
            rf["valu"] = measure_valu_live("c3")
            if "valu_frac" in rf["valu"]:
                rf["valu_frac"] = rf["valu"]["valu_frac"]
                rf["binding"] = ("valu" if rf["valu_frac"] > rf["frac"] else "hbm")
                rf["binding_note"] = ("k_quad420: fraction of the VALU issue bound (valu_frac) against fraction of the HBM bound (frac): the larger one "
                                      "binds; neither is at 1 because the two overlap imperfectly (DESIGN.md 6.1)")
            if "extra" in result and "c4_encode_4096" in result["extra"]:
                c4 = result["extra"]["c4_encode_4096"]
                c4["valu"] = measure_valu_live("c4")
                if "valu_frac" in c4["valu"]:
                    c4["valu_frac"] = c4["valu"]["valu_frac"]
                    c4["binding"] = "valu" if c4["valu_frac"] > c4["frac_hbm"] else "hbm"
        # C5's measured traffic (tools/profile_round.sh -> profiles/traffic_latest.json), reported while it belongs to these kernels
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            key = "c5_%dx1080p" % args.c5_images
            if "c5" in tj and "extra" in result and key in result["extra"] and args.c5_images == tj["c5"].get("images"):
                fresh = tj.get("kernel_source_sha16") == kernel_source_sha16()
                result["extra"][key]["traffic"] = tj["c5"]["hbm_bytes_per_step"] if fresh else None
                result["extra"][key]["traffic_source"] = (f"rocprofv3 PMC passes of the 4096-image job at commit {tj.get('commit')} (tools/profile_round.sh)"
                                                          + ("" if fresh else "; STALE: measured with other kernel sources"))
                if fresh and result["extra"][key].get("GB_per_s"):
                    alg = result["extra"][key]["GB_per_s"] * result["extra"][key]["ms"] * 1e6
                    result["extra"][key]["traffic_over_algorithmic"] = round(tj["c5"]["hbm_bytes_per_step"] / alg, 4)
        except Exception:
            pass
    if rank == 0:
        print(json.dumps(result), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    return result


def run_c5_job(args, make_workload, d_quanta, q_np, rank, world, dev, dist, sync, barrier, check_threads):
    """ONE job of args.c5_images independent 1920x1080 images, contiguous shards (jpeg_amd.dist.shard), every rank its own
    shard, no data-path collective; barrier + synchronize on both sides of the timed steps, max over ranks.  Every rank
    checks one image of its shard (alone == in the batch, == the CPU oracle).  Returns the record on every rank."""
    from jpeg_amd import dist as jd
    lo, hi = jd.shard(args.c5_images, rank, world)
    rec = {"rank": rank, "images": hi - lo}
    # Local work may fail on one rank (e.g. out of memory on a 25-51 GB shard); the COLLECTIVE sequence below -- the two
    # barriers of time_region, max_over_ranks, all_gather_object -- is unconditional, so that a failing rank produces an
    # "error" record instead of pairing its collectives with the wrong ones of the healthy ranks.
    errors, wl = [], None
    try:
        wl = make_workload("c5", 1920, 1080, hi - lo, 1, d_quanta, 20240807 + lo) if hi > lo else None
        if wl is not None:
            wl.step()
        sync()
    except Exception as e:
        errors.append(repr(e)[:200])
        wl = None
    step = wl.step if wl is not None else (lambda: None)
    wall, gpu_ms = time_region(wl, step, args.c5_steps, sync, barrier, errors)
    pixels_all, bytes_all = 1920 * 1080 * args.c5_images, None
    if wl is not None and not errors:
        try:
            rec["gpu_ms_per_step"] = round(gpu_ms / args.c5_steps, 4)
            if not args.no_parity:
                rec.update(wl.verify(q_np, image=(hi - lo - 1) // 2, threads=check_threads))
            bytes_all = wl.bytes // (hi - lo) * args.c5_images
        except Exception as e:
            errors.append(repr(e)[:200])
    del wl
    if errors:
        rec["error"] = "; ".join(errors)
        wall = float("inf")
    else:
        rec["wall_ms_per_step"] = round(wall / args.c5_steps * 1e3, 4)
    wall_max = jd.max_over_ranks(wall if wall != float("inf") else 1e30, dev, dist)
    recs = [rec]
    if dist is not None:
        recs = [None] * world
        dist.all_gather_object(recs, rec)
    ms = wall_max / args.c5_steps * 1e3
    out = {"images": args.c5_images, "n_gpus": world, "steps": args.c5_steps, "scaling": "strong", "ms": round(ms, 4),
           "Mpixels_per_s": round(pixels_all / ms / 1e3, 1), "per_rank": recs,
           "note": "ONE job sharded contiguously over the ranks (jpeg_amd.dist.shard), barrier + max over ranks; the same job at every N"}
    bytes_all = next((b for b in [bytes_all] if b), None)
    if bytes_all:
        out["GB_per_s"] = round(bytes_all / ms / 1e6, 1)
        out["frac_hbm_per_gpu"] = round(bytes_all / ms / 1e6 / HBM_PEAK_GBS / world, 4)
    if any("error" in r for r in recs):
        out["error"] = [r.get("error") for r in recs]
        out.pop("Mpixels_per_s", None)
    checks = [r.get("parity_vs_oracle") for r in recs if r["images"] > 0]
    out["parity_vs_oracle"] = None if (not checks or any(c is None for c in checks)) else bool(all(checks))
    assert sum(r["images"] for r in recs) == args.c5_images
    return out


def main():
    run(parse())


def host_cpus():
    """CPUs this process may keep busy: the affinity mask, capped by the control group's CPU bandwidth (cpu.max): a container
    granted 16 CPUs on a 256-thread host is stopped for the rest of the period when 32 threads run flat out."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def extras(J, ctx, d_quanta, q_np, sync, args, workload="c3"):
    """Short side measurements of the other BASELINE.json configurations (N = 1 only)."""
    import numpy as np
    import torch
    from jpeg_amd import synth, _lib
    out = {}
    dev = ctx.torch_device
    lib = _lib.lib()

    # live device-to-device copy rate (1 GiB, best of 5)
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(6):
        e0.record(); dst.copy_(src); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    out["_d2d_copy_GBps"] = round(2.0 * src.numel() / (best * 1e-3) / 1e9, 1)
    del src, dst

    # C2-shaped: IDCT + dequant only, 2^22 blocks (bandwidth figure; the 100k-block
    # bit-exactness check is tests/test_gpu_parity.py::test_c2_100k_blocks_idct)
    ux, uy = 2048, 2048
    ring = 3
    coef = synth.natural_planes_torch([(ux, uy)], ring, dev, 99)[0]
    plane = torch.empty((ring, 64 * ux * uy), dtype=torch.int16, device=dev)
    q = np.ascontiguousarray(q_np[0])

    def idct_step(i=[0]):
        r = i[0] % ring
        i[0] += 1
        st = lib.jpeg_amd_idct_plane(ctx.handle, coef[r].data_ptr(), ux, uy, q.ctypes.data, 8, plane[r].data_ptr())
        assert st == 0
    for _ in range(5):
        idct_step()
    sync()
    ctx.timer_begin()
    n = 30
    for _ in range(n):
        idct_step()
    ms = ctx.timer_end() / n
    nb = ux * uy
    out["c2_idct_dequant_only"] = {
        "blocks": nb, "ms": round(ms, 4), "Gblocks_per_s": round(nb / ms / 1e6, 2),
        "Mpixels_per_s": round(64 * nb / ms / 1e3, 1),
        "GB_per_s": round(256 * nb / ms / 1e6, 1), "frac_hbm": round(256 * nb / ms / 1e6 / HBM_PEAK_GBS, 4),
        "bytes_per_block": 256}
    del coef, plane

    # C4: encode 4096x4096 RGB8 -> 4:2:0 coefficients (fused pack + decomposed + fdct)
    w = h = 4096
    ring = 3
    px = synth.smooth_rgb_torch(w, h, ring, dev)
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units((w, h))
    L = layout.c_layout((w, h), units, [0, 1, 1])
    coefs = [torch.empty((ring, 64 * a * b), dtype=torch.int16, device=dev) for a, b in units]
    zero = _lib.size_array([0, 0, 0])

    def enc_step(i=[0]):
        r = i[0] % ring
        i[0] += 1
        st = lib.jpeg_amd_encode_batch(ctx.handle, C.byref(L), 1, px[r].data_ptr(), 0, _lib.COLOR_RGB8,
                                       d_quanta.data_ptr(), 0, 2, _lib.ptr_array([c[r].data_ptr() for c in coefs]), zero)
        assert st == 0
    for _ in range(3):
        enc_step()
    sync()
    ctx.timer_begin()
    n = 20
    for _ in range(n):
        enc_step()
    ms = ctx.timer_end() / n
    nbytes = 3 * w * h + 128 * sum(a * b for a, b in units)
    out["c4_encode_4096"] = {"ms": round(ms, 4), "Mpixels_per_s": round(w * h / ms / 1e3, 1),
                             "GB_per_s": round(nbytes / ms / 1e6, 1),
                             "frac_hbm": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
    # the frame just encoded, for the record: PSNR of decode(encode(x)) against x on the device;
    # the coefficient comparison with the CPU restatement happens in the cpu_baseline leg
    # (`_c4_host_case` is handed over to it and removed from the line)
    try:
        st = lib.jpeg_amd_encode_batch(ctx.handle, C.byref(L), 1, px[0].data_ptr(), 0, _lib.COLOR_RGB8,
                                       d_quanta.data_ptr(), 0, 2, _lib.ptr_array([c[0].data_ptr() for c in coefs]), zero)
        assert st == 0
        sync()
        out["_c4_host_case"] = (px[0].cpu().numpy().reshape(-1, 3), [c[0].cpu().numpy() for c in coefs], (w, h))
        back = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
        st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), 1, _lib.ptr_array([c[0].data_ptr() for c in coefs]), zero,
                                       d_quanta.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, back.data_ptr(), 0)
        assert st == 0
        err = (back.view(-1, 3).float() - px[0].view(-1, 3).float()).pow(2).mean().item()
        out["c4_encode_4096"]["roundtrip_psnr_db"] = round(10 * np.log10(255.0 ** 2 / max(err, 1e-12)), 2)
    except Exception as e:  # never let the side measurement break the headline line
        out["c4_encode_4096"]["parity_error"] = repr(e)
    del px, coefs

    # File path, PCIe inclusive (never the headline `value`): 1080p 4:2:0 baseline JPEG bytes in host
    # memory -> RGB bytes in host memory; host threads entropy-decode while the device works on
    # the previous chunk (jpeg_amd_decompress_batch).  The files come from this library's encoder.
    try:
        W, H = 1920, 1080
        yy, xx = np.mgrid[0:H, 0:W]
        rng = np.random.default_rng(5)
        layout = J.Layout("ycc8", {1: ((2, 2), 0), 2: ((1, 1), 1), 3: ((1, 1), 1)})
        quanta = {0: q_np[0], 1: q_np[1]}
        files = []
        for i in range(4):
            base = 128 + 70 * np.sin(xx / (40.0 + 7 * i)) * np.cos(yy / (29.0 + 3 * i))
            rgb = np.clip(base[..., None] + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8).reshape(-1, 3)
            data = J.Rectangular.pack(ctx, (W, H), layout, rgb, J.RGB).compress(
                quanta, [[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]], metadata=[("jfif", (2, 2, 1, 1))])
            files.append(np.frombuffer(data, np.uint8).copy())
        n = 128
        batch = [files[i % 4] for i in range(n)]
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in batch])
        sizes = (C.c_size_t * n)(*[f.size for f in batch])
        pixels = np.empty((n, W * H * 3), np.uint8)
        threads = min(32, host_cpus())
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            st = lib.jpeg_amd_decompress_batch(ctx.handle, ptrs, sizes, n, threads, 0, _lib.COLOR_RGB8, pixels.ctypes.data, 0, None)
            dt = time.perf_counter() - t0
            assert st == 0, st
            best = dt if best is None else min(best, dt)
        out["file_path_1080p_pcie_inclusive"] = {
            "files": n, "host_threads": threads, "ms": round(best * 1e3, 2), "images_per_s": round(n / best, 1),
            "Mpixels_per_s": round(n * W * H / best / 1e6, 1), "jpeg_MB_per_s": round(sum(f.size for f in batch) / best / 1e6, 1),
            "note": "host Huffman decode + H2D + fused decode + D2H; bounded by PCIe and the host, not by the kernels"}
        # the same files with the pixels LEFT ON THE DEVICE (sparse coefficients up, nothing down)
        d_pixels = torch.empty((n, W * H * 3), dtype=torch.uint8, device=ctx.torch_device)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            st = lib.jpeg_amd_decompress_batch_device(ctx.handle, ptrs, sizes, n, threads, 0, _lib.COLOR_RGB8, d_pixels.data_ptr(), 0, None)
            dt = time.perf_counter() - t0
            assert st == 0, st
            best = dt if best is None else min(best, dt)
        out["file_path_1080p_to_device"] = {
            "files": n, "host_threads": threads, "ms": round(best * 1e3, 2), "images_per_s": round(n / best, 1),
            "Mpixels_per_s": round(n * W * H / best / 1e6, 1), "jpeg_MB_per_s": round(sum(f.size for f in batch) / best / 1e6, 1),
            "same_pixels_as_the_host_path": bool((d_pixels.cpu().numpy() == pixels).all()),
            "note": "host Huffman decode into sparse coefficients + H2D + expansion + fused decode; bounded by the host"}
        del d_pixels
        # ... and the other way: the pixels just decoded (host memory) -> baseline JPEG bytes in host memory
        # (jpeg_amd_compress_batch: pinned staging + fused encode + host Huffman coder with optimised tables)
        from jpeg_amd.api import _scan_array, _metadata_array
        info = _lib.FrameInfo()
        info.width, info.height, info.precision, info.ncomponents, info.process = W, H, 8, 3, 0
        for c, (fx, fy) in enumerate([(2, 2), (1, 1), (1, 1)]):
            info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
        tables = np.stack([q_np[0], q_np[1]]).astype(np.uint16)
        qkey, tk = (C.c_int32 * 3)(0, 1, 1), (C.c_int32 * 2)(0, 1)
        sarr = _scan_array([[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]])
        marr, nmeta, _keep = _metadata_array([("jfif", (2, 2, 1, 1))])
        cap = 1 << 20
        jout = np.zeros((n, cap), np.uint8)
        jsizes = (C.c_size_t * n)()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            st = lib.jpeg_amd_compress_batch(ctx.handle, C.byref(info), pixels.ctypes.data, 0, n, _lib.COLOR_RGB8, qkey, tables.ctypes.data,
                                             tk, 2, sarr, 2, marr, nmeta, threads, jout.ctypes.data, cap, jsizes)
            dt = time.perf_counter() - t0
            assert st == 0, st
            best = dt if best is None else min(best, dt)
        out["file_path_1080p_compress_pcie_inclusive"] = {
            "files": n, "host_threads": threads, "ms": round(best * 1e3, 2), "images_per_s": round(n / best, 1),
            "Mpixels_per_s": round(n * W * H / best / 1e6, 1), "jpeg_MB_per_s": round(sum(int(v) for v in jsizes) / best / 1e6, 1),
            "note": "H2D through pinned staging + fused encode + D2H + host Huffman coder (optimised tables); bounded by the host and PCIe"}
    except Exception as e:
        out.setdefault("file_path_1080p_pcie_inclusive", {"error": repr(e)})
        out["file_path_1080p_compress_pcie_inclusive"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    main()
