#!/usr/bin/env python3
"""A/B of a development switch of the fused 4:2:0 decode (default JPEG_AMD_BAND: the band-walk kernel of
kernels_band.hip against the two-launch path): same inputs through both settings (=0 / =1 in two child
processes), output digests compared, times side by side.
Development aid; needs a GPU.  usage: tools/ab_band.py [--quick] [--env=JPEG_AMD_DIRECT]"""
import sys, os, subprocess, json, hashlib, ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(8192, 8192, 1), (4096, 4096, 1), (1920, 1080, 64), (1920, 1080, 512), (2048, 2048, 16), (1000, 700, 40),
         (520, 24, 300), (17, 17, 500), (513, 1030, 20), (8200, 64, 8), (512, 512, 256), (1024, 1024, 1), (24, 4000, 30)]


def child():
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import jpeg_amd as J
    from jpeg_amd import _lib, synth
    ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
    q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
    d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
    cases = CASES[:3] if "--quick" in sys.argv else CASES
    for (W, H, N) in cases:
        layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
        units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
        ring = 3 if W * H * N * 3 < (1 << 30) else 2
        planes = [synth.natural_planes_torch(units, N, dev, 3 + r) for r in range(ring)]
        out = torch.zeros((ring, N * W * H * 3), dtype=torch.uint8, device=dev)
        strides = _lib.size_array([64 * a * b for a, b in units])
        rec = {"case": [W, H, N]}
        for name, color in (("rgb", _lib.COLOR_RGB8), ("ycc", _lib.COLOR_YCC8)):
            def step(i):
                r = i % ring
                st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), N, _lib.ptr_array([p.data_ptr() for p in planes[r]]), strides,
                                               d_q.data_ptr(), 0, 2, 0, color, out[r].data_ptr(), W * H * 3)
                assert st == 0, st
            out.zero_()
            for i in range(ring): step(i)
            torch.cuda.synchronize()
            rec[name + "_sha"] = hashlib.sha1(out[0].cpu().numpy().tobytes()).hexdigest()[:16]
            reps = 30
            ctx.timer_begin()
            for i in range(reps): step(i)
            rec[name + "_us"] = round(ctx.timer_end() / reps * 1e3, 1)
        print(json.dumps(rec), flush=True)


def main():
    res = {}
    var = "JPEG_AMD_BAND"
    for a in sys.argv[1:]:
        if a.startswith("--env="): var = a[6:]
    modes = ("0", "1")
    for a in sys.argv[1:]:
        if a.startswith("--modes="): modes = tuple(a[8:].split(","))
    for mode in modes:
        env = dict(os.environ)
        env[var] = mode
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [a for a in sys.argv[1:]], env=env,
                           capture_output=True, text=True)
        if p.returncode != 0:
            print("child failed (JPEG_AMD_BAND=%s):\n%s\n%s" % (mode, p.stdout[-2000:], p.stderr[-4000:]))
            sys.exit(1)
        res[mode] = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    bad = 0
    for a, b in zip(res[modes[0]], res[modes[1]]):
        same = a["rgb_sha"] == b["rgb_sha"] and a["ycc_sha"] == b["ycc_sha"]
        bad += not same
        W, H, N = a["case"]
        print(f"{N:4d} x {W}x{H}: {var}={modes[0]} RGB {a['rgb_us']:8.1f} YCC {a['ycc_us']:8.1f} us | ={modes[1]} RGB {b['rgb_us']:8.1f} YCC {b['ycc_us']:8.1f} us | "
              f"{'identical' if same else 'DIFFERENT'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
