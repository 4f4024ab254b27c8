#!/usr/bin/env python3
"""VALU issue roofline of the two generic kernels on the 8192 x 8192 4:2:0 12-bit layout (tools/valu_roofline.py; GPU box):
    python tools/valu_generic.py > profiles/rNN_valu_generic.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import valu_roofline as vr
lib = os.path.join(ROOT, "jpeg_amd", "libjpeg_amd.so")
cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_generic.py"), "--only", "4:2:0 12-bit", "--reps", "5"]
out = {}
for name, shown, mangled, waves in (("k_generic_fused", "k_generic_fused<64, 3>", "k_generic_fusedILi64ELi3E", 4),
                                    ("k_generic_encode", "k_generic_encode<3, 32, 256>", "k_generic_encodeILi3ELi32ELi256E", 4)):
    try:
        out[name] = vr.measure(cmd, shown, lib, mangled, waves_per_simd=waves)
    except Exception as e:
        out[name] = {"error": repr(e)[:300]}
print(json.dumps(out, indent=1))
