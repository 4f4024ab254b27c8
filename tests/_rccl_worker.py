"""Worker of tests/test_gpu_rccl_world1.py: a fresh process that initialises torch.distributed with the "nccl" backend
(= RCCL on ROCm) as a world of ONE on cuda:0 -- TCP store on 127.0.0.1, before any other GPU work -- and drives the
multi-GPU plumbing through that group: jpeg_amd.dist.broadcast_quanta (the int32-view broadcast of the uint16 tables),
max_over_ranks (all_reduce MAX), then bench.run() itself with --gpus 1 --dist (barriers, the table broadcast, the C5 job's
all_gather_object).  It proves that the backend loads and that the collectives of the path run on this box; it is NOT a
scaling result.
usage: python _rccl_worker.py <port> <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port, out = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import jpeg_amd as J
    from jpeg_amd import dist as jd

    rec = {}
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    rec["backend"] = dist.get_backend()
    q = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
    t = jd.broadcast_quanta(q, 0, dev, dist)
    torch.cuda.synchronize(dev)
    rec["broadcast_ok"] = bool((t.cpu().numpy().view(np.uint16) == q).all()) and t.dtype == torch.int16
    rec["max_over_ranks"] = jd.max_over_ranks(1.25, dev, dist)
    dist.barrier()
    dist.destroy_process_group()

    import bench
    sys.argv = ["bench.py", "--gpus", "1", "--dist", "--steps", "3", "--warmup", "1", "--c5-images", "6", "--c5-steps", "1",
                "--no-extras", "--no-cpu", "--traffic", "none"]
    rec["result"] = bench.run(bench.parse())
    json.dump(rec, open(out, "w"))


if __name__ == "__main__":
    main()
