#!/usr/bin/env python3
"""Soak of the API surface: contexts created and destroyed, decompress / compress with changing
geometries through the cached staging buffers, free device memory before and after."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib
import _golden as G
lib = _lib.lib()
names = [n for n in G.decode_names(gold_only=True)]
files = {n: np.fromfile(G.path(G.entry(n)["file"]), np.uint8) for n in names}
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
bad = 0
for rep in range(int(os.environ.get('SOAK_REPS', '60'))):
    ctx = J.Context(0, own_stream=bool(rep & 1))
    for k in range(20):
        n = names[(rep * 7 + k) % len(names)]
        e = G.entry(n); f = files[n]
        out = np.empty(e["width"] * e["height"] * 3, np.uint8)
        st = lib.jpeg_amd_decompress(ctx.handle, f.ctypes.data, f.size, 0, J.RGB.code, out.ctypes.data, out.size, None)
        if st != 0 or G.sha(out) != e["gold"]["rgb_sha256"]:
            bad += 1
    ctx.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"{20 * int(os.environ.get('SOAK_REPS', '60'))} decompress calls over {os.environ.get('SOAK_REPS', '60')} contexts in {time.time()-t0:.1f} s, mismatches {bad}, device memory delta {(free0-free1)/1e6:.1f} MB")
