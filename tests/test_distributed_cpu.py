"""world_size-2 gloo test of the multi-GPU plumbing (CPU only; the oracle stands in for the
kernels, which is allowed in tests).  Covers: contiguous image sharding, the table broadcast
from rank 0 (the only collective of the path), max-over-ranks timing reduction, and that the
union of the ranks' outputs equals the single-process result."""
import json
import os
import socket
import subprocess
import sys

import pytest

import _dist_worker as W

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_is_a_partition():
    from jpeg_amd import dist as jd
    for n in (0, 1, 5, 8, 4096):
        for world in (1, 2, 3, 8):
            parts = [jd.shard(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo(tmp_path):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), str(world), str(port), outs[r]])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    tables = W.tables()
    images = W.images()
    want = {str(i): W.decode(images[i], tables) for i in range(W.N_IMAGES)}
    merged = {}
    for o in outs:
        r = json.load(open(o))
        assert r["tables_ok"], "broadcast tables differ from rank 0's"
        assert r["slowest"] == 2.0          # max over ranks of (1 + rank)
        assert sorted(map(int, r["digests"])) == list(range(r["lo"], r["hi"]))
        merged.update(r["digests"])
    assert merged == want
