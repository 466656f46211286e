"""Maximum-size cases: feature matrices with more than 2^31 elements (N . D > 2 147 483 647), i.e. every row address of the
gather / reduce kernels needs 64-bit arithmetic.  The CPU oracle cannot finish these sizes, so parity is checked through
size-independent properties with an exact expected value:

  * all D columns of a row carry the same small integer -> every column of propagate(copy_xj, +) is the SAME exactly
    representable integer sum, which torch's int64 index_add_ over (s mod 1024) gives independently;
  * attention over neighbours that all carry the same value v returns v (softmax weights sum to one), whatever the logits.

A plan holds node ids as int32 and slots / edge positions as unsigned 32-bit values: 2^31 or more NODES and 2^32 - 65536 or more
SLOTS must be REFUSED, not wrapped — and a graph of more than 2^31 edges (COO_T admits any Integer,
GNNGraphs/src/abstracttypes.jl:1) must WORK: the last test builds one and checks it through exact properties."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def _big_graph(torch, n, E, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    s = torch.randint(1, n + 1, (E,), device="cuda", generator=gen, dtype=torch.int64)
    t = torch.randint(1, n + 1, (E,), device="cuda", generator=gen, dtype=torch.int64)
    # make sure the LAST rows (addresses beyond 2^31 elements / 2^33 bytes) are sources AND destinations
    s[:4096] = torch.arange(n - 4095, n + 1, device="cuda")
    t[4096:8192] = torch.arange(n - 4095, n + 1, device="cuda")
    return s, t


@pytest.mark.gpu
@pytest.mark.parametrize("aggr", ["+", "max"])
def test_propagate_beyond_2_pow_31_elements(gm, aggr):
    import torch
    n, D, E = 17_000_000, 128, 30_000_000                 # N . D = 2.18e9 > 2^31
    assert n * D > 2**31
    s, t = _big_graph(torch, n, E, 1)
    g = gm.GNNGraph(s, t, num_nodes=n)
    val = (torch.arange(n, device="cuda", dtype=torch.int64) % 1024)
    x = val.to(torch.float32)[:, None].expand(n, D).contiguous()
    out = gm.propagate(gm.copy_xj, g, aggr, xj=x)
    assert out.shape == (n, D)
    if aggr == "+":
        want = torch.zeros(n, dtype=torch.int64, device="cuda").index_add_(0, t - 1, val[s - 1])
    else:
        want = torch.full((n,), -1, dtype=torch.int64, device="cuda").scatter_reduce_(0, t - 1, val[s - 1], "amax")
    assert int(want.max()) < 2**24
    empty = want < 0 if aggr == "max" else None
    want = want.to(torch.float32)
    if empty is not None:
        want[empty] = float("-inf")                       # empty destinations keep NNlib's identity fill (typemin)
    for c in (0, 1, 63, D - 1):
        assert bool((out[:, c] == want).all()), f"column {c}"
    assert bool((out[n - 4096:] == want[n - 4096:, None]).all())             # the rows past 2^31 elements, every column
    del out, x
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_gather_scatter_beyond_2_pow_31_elements(gm):
    import torch
    from gnnmp.msgpass import _gather
    n, D, K = 17_000_000, 128, 6_000_000
    val = (torch.arange(n, device="cuda", dtype=torch.int64) % 4096)
    x = val.to(torch.float32)[:, None].expand(n, D).contiguous()
    gen = torch.Generator(device="cuda").manual_seed(5)
    idx = torch.randint(n - 2_000_000, n + 1, (K,), device="cuda", generator=gen, dtype=torch.int64)   # rows past 2^31 elements
    y = _gather(x, idx)
    assert y.shape == (K, D)
    want = val[idx - 1].to(torch.float32)
    assert bool((y[:, 0] == want).all()) and bool((y[:, D - 1] == want).all()) and bool((y[:, 77] == want).all())


@pytest.mark.gpu
def test_attention_beyond_2_pow_31_elements(gm):
    import torch
    n, H, C, E = 17_000_000, 8, 16, 24_000_000            # N . H . C = 2.18e9 > 2^31
    gen = torch.Generator(device="cuda").manual_seed(9)
    t = torch.randint(1, n + 1, (E,), device="cuda", generator=gen, dtype=torch.int64)
    t[:4096] = torch.arange(n - 4095, n + 1, device="cuda")
    # sources congruent to the destination mod 7: every neighbour of t (and t itself through its self loop) carries t mod 7
    k = torch.randint(0, (n // 7) - 1, (E,), device="cuda", generator=gen, dtype=torch.int64)
    s = ((t - 1) % 7) + 7 * k + 1
    assert int(s.max()) <= n and int(s.min()) >= 1
    g = gm.GNNGraph(s, t, num_nodes=n)
    val = ((torch.arange(n, device="cuda", dtype=torch.int64) % 7) + 1).to(torch.float32)
    Wx = val[:, None].expand(n, H * C).contiguous()
    # logits that differ per edge (random a): the weights are non-trivial, their sum is still one
    a = torch.randn((2 * C, H), device="cuda", generator=gen) * 0.3
    l = gm.GATConv((4, C), None, heads=H, bias=False, seed=1)
    l.a = a
    from gnnmp import _lib as L
    out = torch.empty((n, H * C), dtype=torch.float32, device="cuda")
    L.check(L.load().gnnmp_gat_conv_f32(g.plan(True).handle, L.ptr(Wx), None, L.ptr(l.a_hc), 0.2, None, L.ACT_IDENTITY,
                                        L.ptr(out), H, C, L.stream_ptr()))
    torch.cuda.synchronize()
    err = (out - val[:, None]).abs().amax(dim=1)
    assert float(err.max()) <= 1e-5 * 7, float(err.max())
    assert float(err[n - 4096:].max()) <= 1e-5 * 7


@pytest.mark.gpu
def test_dense_and_its_adjoints_beyond_2_pow_31_elements(gm):
    import torch
    from gnnmp.backward import dense_grad_w, dense_grad_x
    n, Din, Dout = 17_000_000, 128, 128
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((n, Din), device="cuda", generator=gen)
    W = torch.randn((Dout, Din), device="cuda", generator=gen) / Din ** 0.5
    b = torch.randn((Dout,), device="cuda", generator=gen) * 0.1
    y = gm.dense(x, W, b, "relu")
    tail = slice(n - 8192, n)                              # rows whose addresses lie past 2^31 elements
    ref = torch.relu(x[tail].double() @ W.double().t() + b.double())
    assert float((y[tail].double() - ref).norm()) <= 1e-5 * float(ref.norm())
    head = torch.relu(x[:8192].double() @ W.double().t() + b.double())
    assert float((y[:8192].double() - head).norm()) <= 1e-5 * float(head.norm())
    dx = dense_grad_x(y, W)
    ref = y[tail].double() @ W.double()
    assert float((dx[tail].double() - ref).norm()) <= 1e-5 * float(ref.norm())
    del dx
    # dW = y' x sums over all 17M rows: a miss of the rows past 2^31 elements is an O(1) relative error.  Reference in
    # float64 over row blocks (keeps the float64 copies small).
    dW, db = dense_grad_w(y, x)
    refW = torch.zeros((Dout, Din), dtype=torch.float64, device="cuda")
    refb = torch.zeros((Dout,), dtype=torch.float64, device="cuda")
    for r0 in range(0, n, 1_000_000):
        yb = y[r0:r0 + 1_000_000].double()
        refW += yb.t() @ x[r0:r0 + 1_000_000].double()
        refb += yb.sum(0)
    assert float((dW.double() - refW).norm()) <= 1e-5 * float(refW.norm())
    assert float((db.double() - refb).norm()) <= 1e-5 * float(refb.norm())


@pytest.mark.gpu
def test_plan_refuses_what_int32_cannot_index(gm):
    import torch
    from gnnmp import _lib as L
    lib = L.load()
    s = torch.ones(4, dtype=torch.int64, device="cuda")
    import ctypes
    h = ctypes.c_void_p()
    rc = lib.gnnmp_plan_create(ctypes.byref(h), L.ptr(s), L.ptr(s), 8, 1, 2**31, 2**31, 4, 0, 0, L.stream_ptr())
    assert rc != 0 and not h.value
    rc = lib.gnnmp_plan_create(ctypes.byref(h), L.ptr(s), L.ptr(s), 8, 1, 4, 4, 2**32 + 5, 0, 0, L.stream_ptr())
    assert rc == L.EUNSUPPORTED and not h.value
    rc = lib.gnnmp_plan_create(ctypes.byref(h), L.ptr(s), L.ptr(s), 8, 1, 4, 4, 2**32 - 65536, 0, 0, L.stream_ptr())
    assert rc == L.EUNSUPPORTED and not h.value           # GNNMP_MAX_SLOTS: headroom so that no slot counter can wrap
    rc = lib.gnnmp_plan_create(ctypes.byref(h), L.ptr(s), L.ptr(s), 8, 1, 2**31 - 2, 2**31 - 2, 2**31 + 10, 1, 0, L.stream_ptr())
    assert rc == L.EUNSUPPORTED and not h.value           # E + n self loops does not fit either


@pytest.mark.gpu
def test_graph_with_more_than_2_pow_31_edges(gm):
    """E = 2^31 + 4097 edges on 2^23 nodes (Int32 index arrays: 17 GB; the plan: another 17 GB; mean in-degree 256: rows the
    plan does not split, plus one 100 000-edge hub on the LAST node, whose chunks straddle slot 2^31): slots and edge positions
    pass 2^31, so every signed-32-bit assumption about them would wrap.  Checked through exact properties:
      * rowptr ends at E, row lengths = in-degrees, edge positions are a permutation (checksum) and increase inside a row
        (stability) on a window that straddles slot 2^31;
      * aggregate_neighbors(+) of per-edge integers whose LAST rows (edge positions beyond 2^31) are distinctive = torch's
        int64 index_add_  (the _scatter path: rows addressed by edge position);
      * propagate(copy_xj, +) column = out-degree-weighted integer sum; propagate(w_mul_xj) reads w by edge position;
      * softmax_edge_neighbors sums to one per destination and is invariant to the three-step / one-pass split."""
    import torch
    n, E = 1 << 23, (1 << 31) + 4097
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * (1 << 30):
        pytest.skip("needs ~100 GB of device memory")
    gen = torch.Generator(device="cuda").manual_seed(5)
    s = torch.randint(1, n + 1, (E,), device="cuda", generator=gen, dtype=torch.int32)
    t = torch.randint(1, n + 1, (E,), device="cuda", generator=gen, dtype=torch.int32)
    t[1_000_000:1_100_000] = n                                  # the hub: its slots are the last ones of the plan
    g = gm.GNNGraph(s, t, num_nodes=n)
    p = g.plan(False)
    assert p.n_total == E and p.n_edges == E and p.n_long >= 1
    rowptr, col, eid = p.export64()
    assert int(rowptr[0]) == 0 and int(rowptr[-1]) == E
    indeg = torch.zeros(n, dtype=torch.int64, device="cuda")
    outdeg = torch.zeros(n, dtype=torch.int64, device="cuda")
    CH = 1 << 28
    for a in range(0, E, CH):                                   # chunked: int64 copies of 2^31 indices would be 17 GB each
        indeg += torch.bincount(t[a:a + CH].long() - 1, minlength=n)
        outdeg += torch.bincount(s[a:a + CH].long() - 1, minlength=n)
    assert torch.equal(rowptr[1:] - rowptr[:-1], indeg)
    assert p.max_degree == int(indeg.max())
    tot = 0
    for a in range(0, E, CH):
        tot += int(eid[a:a + CH].sum())
    assert tot == E * (E - 1) // 2                                # a permutation of 0 .. E-1
    assert int(eid.max()) == E - 1 and int(eid.min()) == 0
    # stability + grouping around slot 2^31: inside a row edge positions increase, and every slot's destination is its row
    lo, hi = (1 << 31) - 50_000, E
    row_of = torch.searchsorted(rowptr, torch.arange(lo, hi, device="cuda"), right=True) - 1
    w_eid = eid[lo:hi]
    assert torch.equal(t[w_eid].long() - 1, row_of)
    assert torch.equal(s[w_eid].long() - 1, col[lo:hi].long())
    same = row_of[1:] == row_of[:-1]
    assert bool((w_eid[1:][same] > w_eid[:-1][same]).all())
    del rowptr, col, eid, w_eid, row_of
    torch.cuda.empty_cache()

    # _scatter: one value per edge, the last 4097 edges (positions >= 2^31) carry 1000
    e = torch.ones((E, 1), dtype=torch.float32, device="cuda")
    e[1 << 31:] = 1000.0
    got = gm.aggregate_neighbors(g, "+", e)[:, 0]
    want = indeg.clone()
    want.index_add_(0, t[1 << 31:].long() - 1, torch.full((E - (1 << 31),), 999, dtype=torch.int64, device="cuda"))
    assert int(want.max()) < 2**24
    assert torch.equal(got, want.float())
    # the neighbourhood softmax of the same rows: one pass == three steps, every destination with an edge sums to one
    a1 = gm.softmax_edge_neighbors(g, e)
    gm.tune(16, -1)
    try:
        a3 = gm.softmax_edge_neighbors(g, e)
    finally:
        gm.tune(16, 0)
    assert torch.equal(a1, a3)
    del a3
    sums = torch.zeros(n, dtype=torch.float64, device="cuda")
    for a in range(0, E, CH):
        sums.index_add_(0, t[a:a + CH].long() - 1, a1[a:a + CH, 0].double())
    assert float((sums[indeg > 0] - 1.0).abs().max()) < 1e-4
    del a1, sums
    torch.cuda.empty_cache()

    # propagate: x[j] = j mod 8 in all 4 columns -> exact integer sums
    val = torch.arange(n, device="cuda") % 8
    x = val.float()[:, None].expand(n, 4).contiguous()
    y = gm.propagate(gm.copy_xj, g, "+", xj=x)
    want = torch.zeros(n, dtype=torch.int64, device="cuda")
    for a in range(0, E, CH):
        want.index_add_(0, t[a:a + CH].long() - 1, val[s[a:a + CH].long() - 1])
    assert int(want.max()) < 2**24
    assert torch.equal(y[:, 0], want.float()) and torch.equal(y[:, 3], want.float())
    ymax = gm.propagate(gm.copy_xj, g, "max", xj=x)
    assert float(ymax[indeg > 64].min()) == 7.0                    # 64 uniform draws of 0..7 miss the 7 with probability 2e-4
    # w_mul_xj: weights by edge position, the positions beyond 2^31 weigh 3, the rest 1
    w = e[:, 0].clone()
    w[1 << 31:] = 3.0
    gw = gm.set_edge_weight(g, w)
    yw = gm.propagate(gm.w_mul_xj, gw, "+", xj=x)
    want.index_add_(0, t[1 << 31:].long() - 1, 2 * val[s[1 << 31:].long() - 1])
    assert torch.equal(yw[:, 1], want.float())
    del y, ymax, yw, w, gw, e
    torch.cuda.empty_cache()
    # the one-pass attention kernel and the fused aggregate-then-transform layer walk the same unsigned slots: attention over
    # neighbours that all carry the same row returns that row; the GCN layer equals propagate + dense on the same plan
    l = gm.GATConv((4, 4), None, heads=1, bias=False, add_self_loops=False, seed=3)
    l.dense_x_weight = torch.eye(4, device="cuda")
    v = torch.tensor([0.5, -1.0, 2.0, 0.25], device="cuda")
    ya = l(g, v.repeat(n, 1))
    assert float((ya[indeg > 0] - v[None, :]).abs().max()) < 1e-5
    gm.tune(14, 16)
    try:
        gcn = gm.GCNConv((4, 4), "relu", add_self_loops=False, seed=4)
        yf = gcn(g, x)
    finally:
        gm.tune(14, -1)
    try:
        yu = gcn(g, x)
    finally:
        gm.tune(14, 0)
    assert float((yf - yu).abs().max()) <= 1e-5 * float(yu.abs().max())
