"""1-D destination-row partition of one large graph (gnnmp/rowpart.py; SURVEY.md §8e "next").  CPU: world-size 2 and 3
`gloo` runs whose per-rank compute is the oracle (test-only), required to be BIT-identical to the unpartitioned forward.
GPU: the same partition driven through the HIP kernels rank by rank in one process, bit-identical to the whole-graph
HIP forward."""
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


def _problem(seed=3, n=400, E=6000, D=12):
    rng = np.random.default_rng(seed)
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 7, E)          # the last rows are empty
    t[:900] = 11                           # a hub that unbalances a naive row split
    p = rng.permutation(E)
    s, t = s[p], t[p]
    x = rng.standard_normal((n, D)).astype(np.float32)
    W1 = (rng.standard_normal((D, D)) * 0.3).astype(np.float32)
    W2 = (rng.standard_normal((5, D)) * 0.3).astype(np.float32)
    return s, t, n, x, W1, W2


def _gcn_layer_oracle(orc, s, t, n_src, n_dst, lo, h, c, W, relu):
    """rows [lo, lo + n_dst) of  σ.(W * (c_i Σ_j c_j h_j))  with self loops already in (s, t): conv.jl:43-71 order"""
    m = orc.propagate("+", s, t, n_src, orc.scale_rows(h, c), None, n_dst=n_dst)
    m = orc.scale_rows(m, c[lo:lo + n_dst])
    y = orc.matmul(W, m, blas=False)
    return np.where(y < 0, np.float32(0), y).astype(np.float32) if relu else y


def _full_oracle(orc):
    s, t, n, x, W1, W2 = _problem()
    s2, t2, _ = orc.add_self_loops(s, t, n)
    c = orc.inv_sqrt(orc.degree(t2, n))
    h = _gcn_layer_oracle(orc, s2, t2, n, n, 0, x, c, W1, True)
    return _gcn_layer_oracle(orc, s2, t2, n, n, 0, h, c, W2, False)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from gnnmp import rowpart as RP
        s, t, n, x, W1, W2 = _problem()
        s2, t2, _ = orc.add_self_loops(s, t, n)                     # self loops appended last, as the reference does
        c = orc.inv_sqrt(orc.degree(t2, n))
        bounds = RP.partition_rows_by_edges(torch.from_numpy(t2), n, world)
        lo, hi = bounds[rank]
        sl, tl, keep = RP.local_edges(torch.from_numpy(s2), torch.from_numpy(t2), lo, hi)
        sl, tl = sl.numpy(), tl.numpy()

        def layer(W, relu):
            return lambda h: torch.from_numpy(_gcn_layer_oracle(orc, sl, tl, n, hi - lo, lo, h.numpy(), c, W, relu))

        out = RP.row_parallel_forward([layer(W1, True), layer(W2, False)], torch.from_numpy(x), bounds, rank, world, dist)
        ret[rank] = (out.numpy().copy(), bounds, int(keep.numel()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_parallel_forward_gloo_bit_identical(oracle, world):
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
    full = _full_oracle(oracle)
    edges = 0
    for r in range(world):
        out, bounds, ne = ret[r]
        np.testing.assert_array_equal(out, full)       # a row's edges and their order do not depend on the owner
        edges += ne
    s, t, n, *_ = _problem()
    assert edges == len(s) + n                          # every edge (and self loop) on exactly one rank


def test_partition_balances_edges_not_rows():
    from gnnmp import rowpart as RP
    s, t, n, *_ = _problem()
    tt = torch.from_numpy(t)
    for world in (1, 2, 4, 8):
        b = RP.partition_rows_by_edges(tt, n, world)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        cnt = [int(((tt - 1 >= lo) & (tt - 1 < hi)).sum()) for lo, hi in b]
        assert sum(cnt) == len(t)
        # no rank exceeds its fair share by more than the largest single row (rows are indivisible here)
        assert max(cnt) <= len(t) / world + np.bincount(t).max()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_partition_on_hip_kernels_bit_identical(world):
    import gnnmp
    from gnnmp import _lib as L, rowpart as RP
    from gnnmp.graph import Plan
    gnnmp.load()
    s, t, n, x, W1, _ = _problem(seed=9, n=3000, E=90000, D=100)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    g = gnnmp.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gnnmp.GCNConv((100, 100), "relu", seed=2)
    full = l(g, dev(x))                                               # whole-graph HIP forward (self loops folded in)
    # the partitioned forward: explicit self loops appended last, per-rank bipartite plans
    gl = gnnmp.add_self_loops(g)
    lib = L.load()
    deg = torch.empty(n, dtype=torch.float32, device="cuda")
    L.check(lib.gnnmp_degree_f32(gl.plan(False).handle, None, L.ptr(deg), L.stream_ptr()))
    c = torch.empty_like(deg)
    L.check(lib.gnnmp_inv_sqrt_f32(L.ptr(deg), L.ptr(c), n, L.stream_ptr()))
    bounds = RP.partition_rows_by_edges(gl.t, n, world)
    xd = dev(x)
    outs = []
    for rank in range(world):
        lo, hi = bounds[rank]
        sl, tl, _ = RP.local_edges(gl.s, gl.t, lo, hi)
        plan = Plan(sl, tl, n, hi - lo, 1, False, validate=True)
        agg = torch.empty((hi - lo, 100), dtype=torch.float32, device="cuda")
        cl = c[lo:hi].contiguous()
        L.check(lib.gnnmp_propagate_f32(plan.handle, L.COPY_XJ, L.SUM, L.ptr(xd), None, L.ptr(c), L.ptr(cl), L.ptr(agg), 100,
                                        L.stream_ptr()))
        outs.append(gnnmp.dense(agg, l.weight, l.bias, "relu"))
    part = RP.gather_rows(outs[0], [(0, bounds[0][1])], 0, 1) if world == 1 else torch.cat(outs, 0)
    assert bool((part == full).all())
