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


@pytest.mark.gpu
def test_c_abi_allgather_on_a_one_rank_rccl_communicator():
    """gnnmp_allgather_f32 (csrc/shard.hip) resolves librccl.so with dlopen and calls ncclAllGather on the CALLER's communicator: checked
    here on a communicator of one rank made through ctypes (the box has one GPU) — symbol lookup, argument order, data type code and
    stream are what can go wrong at this boundary, and a world of one exercises all of them; then the whole C-ABI recipe of
    INTEGRATION.md section 2b (shard table, per-rank result, all-gather, gnnmp_gather_f32 by gather_index) for world = 1."""
    import ctypes
    import numpy as np
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graphneuralnetworks.jl_amd"))
    import gnnmp
    from gnnmp import _lib as L
    lib = gnnmp.load()
    torch.cuda.set_device(0)
    rccl = None
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("librccl.so not found on this box")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_byte * 128)]
    uid = UniqueId()
    if rccl.ncclGetUniqueId(ctypes.byref(uid)) != 0:
        pytest.skip("RCCL could not make a unique id on this box (its bootstrap needs a network interface)")
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    if rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) != 0:
        pytest.skip("RCCL could not initialise a one-rank communicator on this box")
    try:
        G, nout = 1000, 2
        sizes = np.random.default_rng(0).integers(20, 41, G).astype(np.int64)
        rank_of = np.empty(G, np.int32); gidx = np.empty(G, np.int64); gmax = ctypes.c_int64()
        L.check(lib.gnnmp_shard_by_size(sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), G, 1,
                                        rank_of.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                        gidx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(gmax)))
        assert gmax.value == G and (rank_of == 0).all() and (gidx == np.arange(G)).all()
        send = torch.randn((gmax.value, nout), device="cuda")
        recv = torch.full((gmax.value, nout), float("nan"), device="cuda")
        L.check(lib.gnnmp_allgather_f32(comm, L.ptr(send), L.ptr(recv), gmax.value * nout, L.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
        out = torch.empty_like(recv)
        idx = torch.from_numpy(gidx).cuda()
        L.check(lib.gnnmp_gather_f32(L.ptr(recv), L.ptr(idx), 8, 0, G, L.ptr(out), nout, L.stream_ptr()))
        assert torch.equal(out, send)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
