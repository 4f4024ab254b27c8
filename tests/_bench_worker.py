"""Worker of tests/test_distributed_cpu.py::test_bench_two_ranks_gloo: one rank of bench.py's own N > 1 code path
(bench.run: sharding with jpeg_amd.dist.shard, table broadcast, barrier + max-over-ranks timing, per-rank records)
under gloo on CPU.  The oracle stands in for the kernels (allowed in tests): the workload object below has the
interface of bench.DecodeWorkload but decodes tiny images on the host.
usage: python _bench_worker.py <rank> <world> <port> <out.json> <c5_images> [fail-rank]
(fail-rank: that rank's C5 workload raises at its first timed step -- the collective sequence must survive it)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402

import _dist_worker as W  # noqa: E402


class OracleWorkload:
    size = W.SIZE

    def __init__(self, name, n_images, quanta, seed, first_image):
        self.name, self.n_images = name, n_images
        self.quanta = quanta.cpu().numpy().view(np.uint16)
        self.first = first_image
        self.images = W.images()
        self.pixels_per_image = W.SIZE[0] * W.SIZE[1]
        self.pixels = self.pixels_per_image * n_images
        self.bytes = 6 * self.pixels
        self.digests = {}
        self._t0 = 0.0

    def step(self):
        for i in range(self.first, self.first + self.n_images):
            self.digests[str(i)] = W.decode(self.images[i % W.N_IMAGES], self.quanta)

    def timer_begin(self):
        self._t0 = time.perf_counter()

    def timer_end(self):
        return (time.perf_counter() - self._t0) * 1e3

    def device_name(self):
        return "cpu (oracle stand-in)"

    def verify(self, quanta_np, image=0, threads=1):
        """Same contract as bench.DecodeWorkload.verify: one image of the shard decoded again on its own and compared."""
        i = self.first + image
        again = W.decode(self.images[i % W.N_IMAGES], self.quanta)
        return {"image_checked": image, "single_equals_batch": (again == self.digests[str(i)]) if self.n_images > 1 else None,
                "parity_vs_oracle": again == self.digests[str(i)]}


def main():
    rank, world, port, out, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    fail_rank = int(sys.argv[6]) if len(sys.argv) > 6 else -1
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import bench
    from jpeg_amd import dist as jd
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--c5-images", str(n), "--c5-steps", "2",
                "--no-extras", "--no-cpu"]
    args = bench.parse()
    made = {}

    def make(name, width, height, n_images, ring, quanta, seed):
        if name == "c3":     # the headline at every N: one image per rank per step (weak scaling)
            assert (width, height, n_images) == (8192, 8192, 1)
            made[name] = OracleWorkload(name, 1, quanta, seed, rank)
        else:                # the collective C5 job leg: ONE job sharded over the ranks
            assert (name, width, height) == ("c5", 1920, 1080)
            lo, hi = jd.shard(n, rank, world)
            assert n_images == hi - lo
            made[name] = OracleWorkload(name, n_images, quanta, seed, lo)
            if rank == fail_rank:   # the set-up step succeeds, the first step inside the timed region raises
                wl, calls = made[name], [0]
                real = wl.step
                def failing_step():
                    calls[0] += 1
                    if calls[0] > 1:
                        raise MemoryError("injected: out of memory on this shard")
                    real()
                wl.step = failing_step
        return made[name]

    result = bench.run(args, make_workload=make, backend="gloo", device_kind="cpu")
    json.dump({"rank": rank, "result": result, "digests": made["c5"].digests, "c3_digests": made["c3"].digests}, open(out, "w"))


if __name__ == "__main__":
    main()
