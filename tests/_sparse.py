"""jpeg_amd_jpeg_decode_sparse for the tests: the call, and the expansion of its entries into planes (numpy)."""
import ctypes as C

import numpy as np


def sparse_decode(lib, data, info, capacity=None):
    blocks = sum(info.units_x[c] * info.units_y[c] for c in range(info.ncomponents))
    desc = np.zeros(blocks, np.uint32)
    ent = np.zeros(capacity if capacity is not None else 64 * blocks + 128, np.uint32)
    quanta = np.zeros((4, 64), np.uint16)
    n = C.c_size_t()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    st = lib.jpeg_amd_jpeg_decode_sparse(buf, len(data), desc.ctypes.data, desc.size, ent.ctypes.data, ent.size, C.byref(n),
                                         quanta.ctypes.data, None)
    return st, desc, ent[:n.value], quanta


def expand(info, desc, ent):
    planes, first = [], 0
    for c in range(info.ncomponents):
        nb = info.units_x[c] * info.units_y[c]
        p = np.zeros((nb, 64), np.int16)
        for b in range(nb):
            s = int(desc[first + b])
            if s == 0xffffffff:
                continue
            while True:
                e = int(ent[s])
                assert (e >> 22) & 0x1ff == 0                      # nothing but value, index and the last-entry flag
                p[b, (e >> 16) & 63] = np.int16(np.uint16(e & 0xffff))
                if e >> 31:
                    break
                s += 1
        planes.append(p.reshape(info.units_y[c], info.units_x[c], 64))
        first += nb
    return planes
