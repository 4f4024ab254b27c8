import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
import jpeg_amd as J
from jpeg_amd import _lib
ctx = J.Context(0); lib = _lib.lib()
fn = lib.jpeg_amd_debug_generic_occupancy; fn.restype = C.c_int; fn.argtypes = [C.POINTER(C.c_int), C.c_int]
out = (C.c_int * 16)()
n = fn(out, 16)
print("k_generic_fused<64,C> C=1..4:", list(out[0:4]), " <32,C>:", list(out[4:8]), " k_generic_encode<C,32,256> (+12 KiB dynamic for C=3):", list(out[8:12]))
