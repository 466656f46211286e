"""N>1 path on CPU: world_size-2 `gloo` run of the graph-parallel forward (shard by graph -> local batch -> forward ->
ONE all-gather of per-shard logits -> original order).  The per-shard forward here is the CPU oracle (this is a test:
the product's forward_local is the HIP one and refuses to run without a GPU); what is under test is the host logic of
gnnmp.parallel — sharding, padding, the collective, the permutation back."""
import os
import socket

import numpy as np
import pytest
import torch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model_oracle(records, weights):
    """GNNChain(GraphConv(16=>32,relu), GraphConv(32=>32,relu), GlobalPool(mean), Dense(32=>2)) on a list of member
    graphs — examples/graph_classification_tudataset.jl:79-82 shape, via the CPU oracle"""
    from oracle import oracle as orc
    if len(records) == 0:
        return torch.zeros((0, 2), dtype=torch.float32)
    s, t, gi, n = orc.batch([(r["s"], r["t"], r["n"]) for r in records])
    x = np.concatenate([r["x"] for r in records])
    W1a, W1b, b1, W2a, W2b, b2, Wd, bd = weights
    h = orc.graph_conv(s, t, n, x, W1a, W1b, b1, "relu", "+", blas=False)
    h = orc.graph_conv(s, t, n, h, W2a, W2b, b2, "relu", "+", blas=False)
    p = orc.global_pool("mean", gi, h, len(records))
    y = orc.matmul(Wd, p, blas=False) + bd[None, :]
    return torch.from_numpy(y.astype(np.float32))


def _make_problem(G=37, seed=5):
    from gnnmp import synth
    rng = np.random.default_rng(seed)
    gs = synth.batched_graphs(G=G, nmin=3, nmax=15, deg=2, seed=seed)
    recs = [{"s": s, "t": t, "n": n, "x": rng.standard_normal((n, 16)).astype(np.float32)} for s, t, n in gs]
    w = [rng.standard_normal(sh).astype(np.float32) * 0.3 for sh in
         [(32, 16), (32, 16), (32,), (32, 32), (32, 32), (32,), (2, 32), (2,)]]
    return recs, w


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnmp.parallel import graph_parallel_forward
        recs, w = _make_problem()
        out, mine = graph_parallel_forward(recs, lambda rs: _model_oracle(rs, w), rank, world, dist,
                                           sizes=[r["n"] for r in recs])
        ret[rank] = (out.numpy().copy(), list(mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_graph_parallel_forward_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    recs, w = _make_problem()
    full = _model_oracle(recs, w).numpy()          # single-process reference: all graphs in one batch
    seen = []
    for r in range(world):
        out, mine = ret[r]
        # graphs are independent units: sharding must not change a single bit of any graph's logits
        np.testing.assert_array_equal(out, full)
        seen += mine
    assert sorted(seen) == list(range(len(recs)))  # a partition: every graph on exactly one rank


def test_shard_by_size_is_balanced_and_deterministic():
    from gnnmp.parallel import shard_by_size
    rng = np.random.default_rng(0)
    sizes = rng.integers(20, 41, size=8192)
    for world in (1, 2, 4, 8):
        sh = shard_by_size(sizes, world)
        assert sh == shard_by_size(sizes, world)
        assert sorted(i for s in sh for i in s) == list(range(len(sizes)))
        counts = [len(s) for s in sh]
        assert max(counts) - min(counts) <= 1
        tot = [int(sizes[s].sum()) for s in sh]
        assert (max(tot) - min(tot)) <= 0.005 * (sum(tot) / world)     # node counts within 0.5 %


def test_gather_without_dist_is_identity_permutation():
    from gnnmp.parallel import gather_shard_outputs, shard_by_size
    sh = shard_by_size([5, 3, 9, 1], 1)
    x = torch.arange(8, dtype=torch.float32).reshape(4, 2)
    assert torch.equal(gather_shard_outputs(x, sh, 0, 1, None), x)


def _worker_plan(rank, world, port, ret):
    """the repeated step: ONE ShardPlan, several gathers with changing logits — no list, no index tensor is rebuilt per step"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnmp.parallel import ShardPlan, shard_by_size
        sizes = [7, 3, 9, 1, 4, 4, 8, 2, 6, 5, 5]
        shards = shard_by_size(sizes, world)
        plan = ShardPlan(shards, rank, world, 2, torch.device("cpu"), torch.float32, dist)
        inv0, send0, recv0 = plan.inv.data_ptr(), plan.send.data_ptr(), plan.recv.data_ptr()
        outs = []
        for step in range(3):
            local = torch.tensor([[100.0 * step + g, -1.0 * g] for g in shards[rank]], dtype=torch.float32).reshape(-1, 2)
            outs.append(plan.gather(local).clone())
        assert (plan.inv.data_ptr(), plan.send.data_ptr(), plan.recv.data_ptr()) == (inv0, send0, recv0)
        ret[rank] = ([o.numpy() for o in outs], plan._flat)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_plan_reused_across_steps(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_plan, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        outs, _flat = ret[r]
        for step, o in enumerate(outs):
            want = np.array([[100.0 * step + g, -1.0 * g] for g in range(11)], np.float32)
            np.testing.assert_array_equal(o, want)


def test_c_abi_shard_table_equals_the_host_mirror():
    """gnnmp_shard_by_size (include/gnnmp.h; host arrays, no GPU needed) deals the member graphs exactly like gnnmp.parallel.shard_by_size,
    and its gather_index is the inverse permutation ShardPlan builds"""
    import ctypes
    from gnnmp import _lib as L
    from gnnmp.parallel import ShardPlan, shard_by_size
    lib = L.load()
    rng = np.random.default_rng(3)
    for G, world in ((0, 2), (1, 1), (5, 8), (37, 3), (8192, 8), (1000, 7)):
        sizes = rng.integers(1, 60, size=G).astype(np.int64)
        rank_of = (ctypes.c_int32 * max(G, 1))()
        gidx = (ctypes.c_int64 * max(G, 1))()
        gmax = ctypes.c_int64(-1)
        L.check(lib.gnnmp_shard_by_size((ctypes.c_int64 * max(G, 1))(*sizes.tolist()), G, world, rank_of, gidx, ctypes.byref(gmax)))
        shards = shard_by_size(sizes.tolist(), world)
        want_rank = np.empty(G, np.int32)
        for r, s in enumerate(shards):
            want_rank[s] = r
        np.testing.assert_array_equal(np.array(rank_of[:G], np.int32), want_rank)
        plan = ShardPlan(shards, 0, world, 2, torch.device("cpu"))
        assert gmax.value == plan.gmax
        np.testing.assert_array_equal(np.array(gidx[:G], np.int64), plan.inv.numpy())
    assert lib.gnnmp_shard_by_size(None, 3, 2, None, None, None) == L.EINVAL


def _oracle_forward_factory(members, xs):
    """bench.batched_setup's forward on the CPU: config 5's chain through the oracle with fixed weights"""
    from oracle import oracle as orc
    rng = np.random.default_rng(77)
    W = [rng.standard_normal(sh).astype(np.float32) * 0.2 for sh in [(128, 16), (128, 16), (128,), (128, 128), (128, 128), (128,), (2, 128), (2,)]]

    def forward():
        if not members:
            return torch.zeros((0, 2), dtype=torch.float32)
        s, t, gi, n = orc.batch(members)
        h = orc.graph_conv(s, t, n, np.concatenate(xs), W[0], W[1], W[2], "relu", "+", blas=False)
        h = orc.graph_conv(s, t, n, h, W[3], W[4], W[5], "relu", "+", blas=False)
        return torch.from_numpy((orc.matmul(W[6], orc.global_pool("mean", gi, h, len(members)), blas=False) + W[7][None, :]).astype(np.float32))
    return forward


def _worker_bench(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        step, G, n_tot, e_tot = bench.batched_setup(rank, world, dist, G=96, forward_factory=_oracle_forward_factory, device=torch.device("cpu"))
        a = step().clone()
        b = step().clone()
        assert torch.equal(a, b) and a.shape == (G, 2)
        # the driver's timing protocol exactly as bench.py runs it for extras.batched_strong / --workload batched: warm-up, barrier, K
        # steps, barrier, MAX over the ranks.  Rank 1 is made slow: every rank must report ITS time, i.e. the maximum.
        import time
        calls = [0]

        def slow_step():
            calls[0] += 1
            if rank == 1:
                time.sleep(0.02)
            return step()
        dt, out = bench.timed_region(slow_step, 3, 2, dist.barrier, dist, device="cpu")
        assert calls[0] == 5 and torch.equal(out, a)
        assert dt >= 3 * 0.02
        ret[rank] = (a.numpy(), dt)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_batched_strong_code_path_gloo(world):
    """the step bench.py times for extras.batched_strong / --workload batched (batched_setup: shard, forward, ShardPlan.gather) and the
    timing protocol around it (bench.timed_region), dry-run on the CPU over gloo with the oracle as the forward — at world 2 and at the
    world 8 of the driver's scaling run: every rank ends with the unsharded logits, bit for bit, and with the same (maximum) time"""
    import sys
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    import bench
    step, G, _, _ = bench.batched_setup(0, 1, None, G=96, forward_factory=_oracle_forward_factory, device=torch.device("cpu"))
    full = step().numpy()
    for r in range(world):
        np.testing.assert_array_equal(ret[r][0], full)
        assert ret[r][1] == ret[0][1]                  # the all-reduced maximum, identical on every rank
