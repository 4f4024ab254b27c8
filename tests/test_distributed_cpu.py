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


@pytest.mark.parametrize("world,n", [(2, 5), (8, 4096)], ids=["world2", "world8_4096_images"])
def test_bench_two_ranks_gloo(tmp_path, world, n):
    """bench.py's own N > 1 path (bench.run) under gloo, the oracle standing in for the kernels: world size 2, and world
    size 8 with BASELINE.json configs[4]'s 4096 images -- the exact rank / shard arithmetic of the 8-GPU run (512 images
    per rank), executed somewhere before the driver's node does it.
    `value` stays on ONE workload at every N (C3: an image per rank per step, weak scaling); the sharded C5 job is
    measured collectively at every N and reported as extra.c5_<n>x1080p (strong scaling) -- every image decoded by
    exactly one rank; every rank proves what it timed (per_rank[i].parity_vs_oracle) for both."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = [str(tmp_path / f"bench{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_bench_worker.py"), str(r), str(world), str(port), outs[r], str(n)],
                              stdout=subprocess.PIPE, text=True) for r in range(world)]
    lines = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0
        lines += [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    line = json.loads(lines[0])
    recs = [json.load(open(o)) for o in outs]
    assert all(r["result"] is None for r in recs[1:]) and recs[0]["result"] == line
    # the headline: the same per-GPU work at every N
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["steps"] == 2
    assert line["config"]["workload"].startswith("C3") and line["config"]["images_per_step_all_ranks"] == world
    assert [r["rank"] for r in line["per_rank"]] == list(range(world))
    assert all(r["images_per_step"] == 1 for r in line["per_rank"])
    assert all(r["parity_vs_oracle"] is True for r in line["per_rank"]) and line["parity_vs_oracle"] is True
    slowest = max(r["wall_ms_per_step"] for r in line["per_rank"])
    assert abs(line["ms_per_step"] - slowest) < 1e-3 * slowest + 1e-3
    px = W.SIZE[0] * W.SIZE[1] * world
    assert abs(line["value"] - px / (line["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * line["value"] + 0.1
    assert "extra.c5_%dx1080p" % n in line["scaling_curve"]
    # the sharded C5 job at this N: one record, all ranks, every rank self-checked
    job = line["extra"]["c5_%dx1080p" % n]
    assert job["n_gpus"] == world and job["scaling"] == "strong" and job["images"] == n and "error" not in job
    assert [r["rank"] for r in job["per_rank"]] == list(range(world)) and sum(r["images"] for r in job["per_rank"]) == n
    if n % world == 0:
        assert all(r["images"] == n // world for r in job["per_rank"])   # 4096 over 8: 512 each, contiguous (jpeg_amd.dist.shard)
    assert all(r["parity_vs_oracle"] is True and r["single_equals_batch"] is True for r in job["per_rank"])
    assert job["parity_vs_oracle"] is True
    slowest = max(r["wall_ms_per_step"] for r in job["per_rank"])
    assert abs(job["ms"] - slowest) < 1e-3 * slowest + 1e-3
    assert abs(job["Mpixels_per_s"] - 1920 * 1080 * n / job["ms"] / 1e3) <= 0.01 * job["Mpixels_per_s"] + 0.1
    tables = W.tables()
    images = W.images()
    merged = {}
    for r in recs:
        assert not (set(merged) & set(r["digests"])), "an image was decoded by two ranks"
        merged.update(r["digests"])
    assert merged == {str(i): W.decode(images[i % W.N_IMAGES], tables) for i in range(n)}


def test_bench_two_ranks_one_failing_c5_shard_keeps_the_collectives_paired(tmp_path):
    """a rank whose C5 shard throws inside the timed region must still meet both barriers, max_over_ranks and
    all_gather_object -- the job then carries an "error" record (and no throughput) instead of hanging or pairing its
    collectives with the wrong ones of the healthy rank; the headline (C3) is untouched."""
    world, n = 2, 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = [str(tmp_path / f"bench{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_bench_worker.py"), str(r), str(world), str(port), outs[r], str(n), "1"],
                              stdout=subprocess.PIPE, text=True) for r in range(world)]
    lines = []
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0
        lines += [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["parity_vs_oracle"] is True and line["value"] > 0
    job = line["extra"]["c5_%dx1080p" % n]
    assert "Mpixels_per_s" not in job and job["error"][0] is None and "injected" in job["error"][1]
    assert [r["rank"] for r in job["per_rank"]] == [0, 1] and "error" not in job["per_rank"][0]
    assert job["per_rank"][0]["parity_vs_oracle"] is True


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("size,factors", [((200, 333), [(2, 2), (1, 1), (1, 1)]), ((96, 64), [(2, 1), (1, 1), (1, 1)]),
                                           ((50, 90), [(1, 1), (1, 1), (1, 1)])])
def test_bands_of_one_image_equal_the_whole_decode(world, size, factors):
    """jpeg_amd.dist.band: every rank decodes its MCU rows + one halo MCU row as an independent
    sub-image (the oracle stands in for the kernels); stacking the kept rows gives the whole-image
    decode bit for bit."""
    import numpy as np
    from jpeg_amd import dist as jd
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    scale = (max(f[0] for f in factors), max(f[1] for f in factors))
    units = [O.plane_units(size, f, scale) for f in factors]
    planes = []
    for ux, uy in units:
        c = rng.integers(-200, 200, (uy, ux, 64)).astype(np.int16)
        c[..., 10:] //= 16
        planes.append(c)
    quanta = [rng.integers(1, 30, 64).astype(np.uint16) for _ in factors]
    _, rect = O.decode(planes, quanta, factors, size)
    whole = O.unpack_rgb8(rect, 3).reshape(size[1], size[0], 3)
    rows = []
    for rank in range(world):
        plan = jd.band(size, scale, rank, world)
        if plan is None:
            continue
        sub = []
        for p, (f, (ux, uy)) in zip(planes, zip(factors, units)):
            u0, u1 = jd.band_units(plan, f[1], uy)
            sub.append(np.ascontiguousarray(p[u0:u1]))
        _, r = O.decode(sub, quanta, factors, (size[0], plan["height"]))
        px = O.unpack_rgb8(r, 3).reshape(plan["height"], size[0], 3)
        y0, y1 = plan["rows"]
        rows.append(px[plan["skip"]:plan["skip"] + (y1 - y0)])
    assert (np.concatenate(rows) == whole).all()
