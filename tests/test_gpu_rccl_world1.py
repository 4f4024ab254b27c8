"""RCCL once: the "nccl" backend of torch.distributed -- RCCL on ROCm -- initialised as a world of one
on the GPU box, in a fresh child process (a child is started, nothing is re-executed), and the path's collectives driven
through it: the table broadcast, the max over ranks, bench.run() with --gpus 1 --dist.  Not a scaling result: it shows that
the int32-view broadcast and the backend work on this box, which no round had executed before."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_rccl_world_of_one_runs_the_paths_collectives_and_the_bench(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_worker.py"), str(port), out], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["backend"] == "nccl"
    assert rec["broadcast_ok"] is True and rec["max_over_ranks"] == 1.25
    line = rec["result"]
    assert [l for l in p.stdout.splitlines() if l.startswith("{")] == [json.dumps(line)], "exactly one JSON line"
    assert line["n_gpus"] == 1 and line["parity_vs_oracle"] is True and line["value"] > 0
    job = line["extra"]["c5_6x1080p"]
    assert "error" not in job and job["parity_vs_oracle"] is True and job["per_rank"][0]["single_equals_batch"] is True
