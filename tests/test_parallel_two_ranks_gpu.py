"""The graph-parallel step of BASELINE.json config 5 with two real processes running the HIP kernels: bench.py's own
batched_setup on each rank's shard, gnnmp.parallel.ShardPlan's device-side exchange, logits identical on both ranks and bit-identical
to the unsharded step.  Both ranks share the box's single GPU and talk over gloo (RCCL wants a device per rank; the collective's
backend is the one thing this test cannot exercise — the driver's multi-GPU run does)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("G", [257, 2048])
def test_two_ranks_on_one_gpu_match_the_unsharded_step(G):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "two_ranks_one_gpu.py"), str(G)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert f"TWO_RANKS_OK G={G}" in r.stdout, r.stdout[-2000:]
