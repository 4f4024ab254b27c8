"""Worker of tests/test_distributed_cpu.py: one rank of a gloo process group (CPU).
usage: python _dist_worker.py <rank> <world> <port> <out.json>"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_IMAGES, SIZE = 5, (48, 40)
FACTORS = [(2, 2), (1, 1), (1, 1)]


def images():
    from jpeg_amd import synth
    from oracle import oracle as O
    units = [O.plane_units(SIZE, f, (2, 2)) for f in FACTORS]
    return [[synth.blocks_natural(ux * uy, 100 * i + p).reshape(uy, ux, 64) for p, (ux, uy) in enumerate(units)]
            for i in range(N_IMAGES)]


def decode(planes, quanta):
    from oracle import oracle as O
    _, rect = O.decode(planes, [quanta[0], quanta[1], quanta[1]], FACTORS, SIZE)
    return hashlib.sha256(O.unpack_rgb8(rect, 3).tobytes()).hexdigest()


def tables():
    import jpeg_amd
    return np.stack([jpeg_amd.compression_quanta("luminance", 1.0), jpeg_amd.compression_quanta("chrominance", 1.0)])


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jpeg_amd import dist as jd
        t = jd.broadcast_quanta(tables() if rank == 0 else 2, 0, torch.device("cpu"), dist)
        got = t.numpy().view(np.uint16)
        lo, hi = jd.shard(N_IMAGES, rank, world)
        imgs = images()
        digests = {str(i): decode(imgs[i], got) for i in range(lo, hi)}
        slowest = jd.max_over_ranks(1.0 + rank, torch.device("cpu"), dist)
        json.dump({"rank": rank, "lo": lo, "hi": hi, "digests": digests, "slowest": slowest,
                   "tables_ok": bool((got == tables()).all())}, open(out, "w"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
