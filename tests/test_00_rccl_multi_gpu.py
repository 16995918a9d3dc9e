"""The RCCL paths (backend "nccl") of parallel.allreduce_gradients and parallel.ShardedUpdate on real devices.

Collected first on purpose: when it runs, this pytest process has made no GPU call yet (counting devices makes
none), and the ranks are started as fresh child processes (python -m torch.distributed.run, 127.0.0.1), one per GPU.
Skipped with fewer than two visible GPUs; tests/dist_worker.py is also run with gloo on the CPU by
test_parallel_gloo.py, so everything but the RCCL transport is rehearsed without a second device."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")
def test_allreduce_and_sharded_update_over_rccl():
    world = min(torch.cuda.device_count(), 8)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "--backend", "nccl"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for r in range(world):
        assert f"rank {r} ok" in out.stdout
