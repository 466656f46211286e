"""Float64 features at the seam (round 6; include/gnnmp.h: gnnmp_propagate_f64 / gnnmp_gather_f64 / gnnmp_scatter_f64).  The reference's
message passing is eltype-generic (GNNlib/src/msgpass.jl:71-79, GNNGraphs/src/gatherscatter.jl:4,12-18) and its own micro-benchmark runs
in Float64 and asserts isequal(propagate(e_mul_xj, g, +; xj = B, e), B * A) (GraphNeuralNetworks/perf/bench_gnn.jl:9-40).  Same bars as
Float32: bit-identical to the CPU gather -> scatter loop on every destination the plan does not split, split rows deterministic and within
1e-12 here (double), identities 0 / -Inf / +Inf on empty destinations."""
import numpy as np
import pytest


def test_oracle_float64_loops_follow_numpy_unbuffered_adds(oracle):
    """the Float64 restatement against numpy's unbuffered, index-ordered ufunc.at — an independent sequential scatter (no GPU)"""
    rng = np.random.default_rng(0)
    n, E, D = 50, 700, 7
    s, t = rng.integers(1, n + 1, E), rng.integers(1, n + 1, E)
    x = rng.standard_normal((n, D))
    w = rng.random(E)
    ref = np.zeros((n, D))
    np.add.at(ref, t - 1, w[:, None] * x[s - 1])
    assert np.array_equal(oracle.propagate("+", s, t, n, x, w), ref)
    mx = np.full((n, D), -np.inf)
    np.maximum.at(mx, t - 1, x[s - 1])
    assert np.array_equal(oracle.propagate("max", s, t, n, x), mx)
    cnt = np.bincount(t - 1, minlength=n)[:, None]
    sm = np.zeros((n, D))
    np.add.at(sm, t - 1, x[s - 1])
    assert np.array_equal(oracle.propagate("mean", s, t, n, x), 0.0 + np.where(cnt == 0, sm, sm / np.maximum(cnt, 1)))
    assert oracle.gather(x, s).dtype == np.float64 and np.array_equal(oracle.gather(x, s), x[s - 1])
    assert oracle.propagate("+", s, t, n, x.astype(np.float32)).dtype == np.float32      # Float32 stays Float32


def _hub_graph(rng, n, m, hubs):
    s = [rng.integers(0, n, m)]
    t = [rng.integers(0, n, m)]
    for node, deg in hubs:
        s.append(rng.integers(0, n, deg))
        t.append(np.full(deg, node))
    s, t = np.concatenate(s).astype(np.int64), np.concatenate(t).astype(np.int64)
    p = rng.permutation(len(s))
    return s[p] + 1, t[p] + 1


@pytest.mark.gpu
@pytest.mark.parametrize("D", [1, 3, 10, 100, 130])
def test_propagate_f64_matches_the_cpu_loop(oracle, D):
    import torch
    import gnnmp as gm
    gm.load()
    rng = np.random.default_rng(D)
    n = 2500
    s, t = _hub_graph(rng, n - 30, 30000, [(5, 70), (9, 700), (11, 5000)])      # split rows (threshold 64) + 30 isolated nodes
    x = rng.standard_normal((n, D))
    w = rng.random(len(s)) + 0.1
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), torch.from_numpy(w.astype(np.float32)).cuda(), num_nodes=n)
    xd = torch.from_numpy(x).cuda()
    assert xd.dtype == torch.float64
    short = np.bincount(t - 1, minlength=n) <= g.plan(False).long_thresh
    assert (~short).sum() >= 3
    for aggr in ("+", "mean", "max", "min"):
        y = gm.propagate(gm.copy_xj, g, aggr, xj=xd)
        assert y.dtype == torch.float64
        y = y.cpu().numpy()
        ref = oracle.propagate(aggr, s, t, n, x)
        assert np.array_equal(y[short], ref[short]), f"copy_xj, {aggr}: unsplit rows differ from the CPU loop"
        if aggr in ("max", "min"):
            assert np.array_equal(y, ref)
        else:
            assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
        # twice: same bits
        assert torch.equal(gm.propagate(gm.copy_xj, g, aggr, xj=xd), torch.from_numpy(y).cuda())
    # graph weights (Float32 in the graph, promoted like `w .* xj`) and an explicit Float64 vector e
    w32 = w.astype(np.float32).astype(np.float64)
    y = gm.propagate(gm.w_mul_xj, g, "+", xj=xd).cpu().numpy()
    ref = oracle.propagate("+", s, t, n, x, w32)
    assert np.array_equal(y[short], ref[short]) and np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    e = rng.standard_normal(len(s))
    y = gm.propagate(gm.e_mul_xj, g, "mean", xj=xd, e=torch.from_numpy(e).cuda()).cpu().numpy()
    ref = oracle.propagate("mean", s, t, n, x, e)
    assert np.array_equal(y[short], ref[short]) and np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    # Float32 features with a Float64 e: Julia's `e .* xj` promotes
    y = gm.propagate(gm.e_mul_xj, g, "+", xj=xd.to(torch.float32), e=torch.from_numpy(e).cuda())
    assert y.dtype == torch.float64


@pytest.mark.gpu
def test_reference_microbenchmark_assertion_in_float64(oracle):
    """GraphNeuralNetworks/perf/bench_gnn.jl:7-40: A = sprand(n, n, 0.01), B = rand(100, n), g = GNNGraph(A; graph_type = :coo);
    @assert isequal(propagate(e_mul_xj, g, +; xj = B, e = A.nzval), B * A) — bit equality with the dense x CSC product (for col, for k in
    nzrange: C[:, col] += B[:, row_k] * val_k, multiply and add rounded separately), here against that loop restated in numpy"""
    import scipy.sparse as sp
    import torch
    import gnnmp as gm
    gm.load()
    rng = np.random.default_rng(0)
    n = 1024
    A = sp.random(n, n, density=0.01, format="csc", random_state=rng, dtype=np.float64)
    A.sort_indices()
    B = rng.random((n, 100))                                   # (100, n) column-major == [n][100]
    # findnz(A): column by column, rows ascending — the COO graph the reference builds from A
    t = np.repeat(np.arange(1, n + 1), np.diff(A.indptr)).astype(np.int64)
    s = (A.indices + 1).astype(np.int64)
    e = A.data.copy()
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=n)
    y = gm.propagate(gm.e_mul_xj, g, "+", xj=torch.from_numpy(B).cuda(), e=torch.from_numpy(e).cuda()).cpu().numpy()
    ref = np.zeros((n, 100))
    for col in range(n):
        for k in range(A.indptr[col], A.indptr[col + 1]):
            ref[col] = ref[col] + B[A.indices[k]] * A.data[k]
    assert np.array_equal(y, ref), "isequal(propagate(e_mul_xj, g, +), B * A) does not hold in Float64"
    # and the closure spelling of the same benchmark, (xi, xj, e) -> e .* xj, through the generic gather -> f -> scatter path
    y2 = gm.propagate(lambda xi, xj, ee: ee * xj, g, "+", xj=torch.from_numpy(B).cuda(), e=torch.from_numpy(e).cuda().reshape(-1, 1))
    assert y2.dtype == torch.float64 and np.array_equal(y2.cpu().numpy(), ref)


@pytest.mark.gpu
def test_leaves_f64(oracle):
    import torch
    import gnnmp as gm
    from gnnmp import msgpass
    gm.load()
    rng = np.random.default_rng(3)
    n, K, D = 300, 4000, 6
    x = rng.standard_normal((n, 2, 3))                         # trailing dims are kept
    idx = rng.integers(1, n + 1, K)
    got = msgpass._gather(torch.from_numpy(x).cuda(), torch.from_numpy(idx).cuda())
    assert got.dtype == torch.float64 and np.array_equal(got.cpu().numpy(), x[idx - 1])
    m = rng.standard_normal((K, D))
    for aggr in ("+", "mean", "max", "min"):
        got = msgpass._scatter(aggr, torch.from_numpy(m).cuda(), torch.from_numpy(idx).cuda(), n)
        ref = oracle.scatter(aggr, m, idx, n)
        assert got.dtype == torch.float64 and np.array_equal(got.cpu().numpy(), ref), aggr
    # empty graph / zero width
    g = gm.GNNGraph(torch.zeros(0, dtype=torch.int64).cuda(), torch.zeros(0, dtype=torch.int64).cuda(), num_nodes=5)
    y = gm.propagate(gm.copy_xj, g, "max", xj=torch.from_numpy(rng.standard_normal((5, 4))).cuda())
    assert y.dtype == torch.float64 and torch.isinf(y).all() and (y < 0).all()
    # the layers' fused kernels stay Float32: a Float64 feature matrix is refused loudly, never converted silently
    gcn = gm.GCNConv((4, 4))
    g2 = gm.GNNGraph(torch.tensor([1, 2]).cuda(), torch.tensor([2, 1]).cuda(), num_nodes=5)
    with pytest.raises(AssertionError):
        gcn(g2, torch.from_numpy(rng.standard_normal((5, 4))).cuda())


@pytest.mark.gpu
def test_reduce_nodes_f64(oracle):
    """reduce_nodes(aggr, g, x) = scatter(aggr, x, graph_indicator) (GNNlib/src/utils.jl:12-16) for Float64 node features: the Float64
    scatter over a plan of the indicator, rows in node order — bit-identical to the CPU loop (segments of a batch are short rows)"""
    import torch
    import gnnmp as gm
    from gnnmp import synth
    gm.load()
    members = synth.batched_graphs(G=40, seed=2)
    rng = np.random.default_rng(5)
    xs = [rng.standard_normal((n, 5)).astype(np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    x64 = torch.from_numpy(rng.standard_normal((g.num_nodes, 5))).cuda()
    gi = g.graph_indicator.cpu().numpy()
    for aggr in ("+", "mean", "max", "min"):
        got = gm.reduce_nodes(aggr, g, x64)
        assert got.dtype == torch.float64
        assert np.array_equal(got.cpu().numpy(), oracle.scatter(aggr, x64.cpu().numpy(), gi, g.num_graphs)), aggr


@pytest.mark.gpu
def test_max_min_special_values_f64(oracle):
    """Base.max / Base.min in Float64: first NaN met wins (payload included), -0.0 < +0.0, +-Inf ordinary — every bit against the oracle"""
    import torch
    import gnnmp as gm
    gm.load()
    rng = np.random.default_rng(11)
    n, E, D = 300, 5000, 6
    s = rng.integers(1, n + 1, E).astype(np.int64)
    t = rng.integers(1, n - 10 + 1, E).astype(np.int64)
    x = rng.standard_normal((n, D))
    x[rng.random((n, D)) < 0.3] = 0.0
    x[rng.random((n, D)) < 0.15] = -0.0
    specials = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-310, -1e-310, 5e-324])      # (with subnormals)
    pick = rng.random((n, D)) < 0.1
    x[pick] = specials[rng.integers(0, len(specials), int(pick.sum()))]
    bits = x.view(np.uint64)
    isn = np.isnan(x)
    bits[isn] = np.where(rng.random(int(isn.sum())) < 0.5, 0x7FF8000000000001, 0xFFF8000000000123).astype(np.uint64)
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=n)
    for aggr in ("max", "min"):
        got = gm.propagate(gm.copy_xj, g, aggr, xj=torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.array_equal(got.view(np.uint64), oracle.propagate(aggr, s, t, n, x).view(np.uint64)), aggr
