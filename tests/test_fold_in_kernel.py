"""Split rows folded INSIDE the row kernels (round 5; csrc/csr_reduce.h long_geom / chunk_arrive, csrc/gat_fused.hip gat_fold_slice):
the last chunk of a slice / the last slice of a long row to arrive does what csr_combine_kernel / gat_fused_combine_kernel did in a second
launch — the same operations in the same order, so every output must be BIT-IDENTICAL to the two-kernel path (knob 19 bit 7), on the first
launch and on every later one (the arrival counters must come back to zero), for every aggregation, with and without edge weights, for
rows of one and of several feature tiles, and for hubs of one chunk per slice up to hundreds of chunks.  Reference semantics of the rows
themselves: GNNlib/src/msgpass.jl:71-79,145-149 (propagate), GNNlib/src/layers/conv.jl:136-141 (GATConv) — checked against the oracle by the
parity suites, which run through the folding kernels by default."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def hub_graph(rng, n, m, hubs):
    """random multigraph + hubs: (destination, in-degree) pairs, so rows of 1 .. hundreds of chunks exist"""
    s = [rng.integers(0, n, m)]
    t = [rng.integers(0, n, m)]
    for node, deg in hubs:
        s.append(rng.integers(0, n, deg))
        t.append(np.full(deg, node))
    s = np.concatenate(s).astype(np.int64)
    t = np.concatenate(t).astype(np.int64)
    perm = rng.permutation(len(s))
    return s[perm] + 1, t[perm] + 1


HUBS = [(3, 65), (10, 128), (11, 129), (77, 513), (200, 1500), (201, 4100), (500, 20000)]


def both(gm, fn):
    """fn() under the folding kernels (twice: the second launch reuses the counters) and under the two-kernel path"""
    import torch
    before = gm.knob(19)
    try:
        gm.tune(19, before & ~128)
        a1 = fn()
        a2 = fn()
        gm.tune(19, before | 128)
        b = fn()
    finally:
        gm.tune(19, before)
    torch.cuda.synchronize()
    return a1, a2, b


@pytest.mark.parametrize("D", [4, 100, 128, 300, 1000])
@pytest.mark.parametrize("aggr", ["+", "mean", "max", "min"])
def test_propagate_fold_is_bit_identical_to_the_combine_kernel(gm, D, aggr):
    import torch
    rng = np.random.default_rng(D * 7 + len(aggr))
    n = 3000
    s, t = hub_graph(rng, n, 20000, HUBS)
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=n)
    assert g.plan().n_long >= len(HUBS)
    x = torch.from_numpy(rng.standard_normal((n, D), dtype=np.float32)).cuda()
    w = torch.from_numpy(rng.standard_normal(len(s)).astype(np.float32)).cuda()
    a1, a2, b = both(gm, lambda: gm.propagate(gm.copy_xj, g, aggr, xj=x))
    assert torch.equal(a1.view(torch.int32), b.view(torch.int32)) and torch.equal(a2.view(torch.int32), b.view(torch.int32))
    if aggr in ("+", "mean"):
        a1, a2, b = both(gm, lambda: gm.propagate(gm.e_mul_xj, g, aggr, xj=x, e=w))
        assert torch.equal(a1.view(torch.int32), b.view(torch.int32)) and torch.equal(a2.view(torch.int32), b.view(torch.int32))


def test_scatter_and_degree_paths(gm):
    """_scatter (idx = edge position) and the GCN layer (slot-ordered coefficients, destination scaling, bias + relu epilogue) go through
    the same kernel with other flags"""
    import torch
    rng = np.random.default_rng(3)
    n = 2000
    s, t = hub_graph(rng, n, 9000, HUBS[:5])
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=n)
    m = torch.from_numpy(rng.standard_normal((len(s), 12), dtype=np.float32)).cuda()
    for aggr in ("+", "mean", "max", "min"):
        a1, a2, b = both(gm, lambda: gm.aggregate_neighbors(g, aggr, m))
        assert torch.equal(a1.view(torch.int32), b.view(torch.int32)) and torch.equal(a2.view(torch.int32), b.view(torch.int32))
    x = torch.from_numpy(rng.standard_normal((n, 64), dtype=np.float32)).cuda()
    for dims in ((64, 64), (64, 16)):          # aggregate-first and W-first (the row kernel's bias + relu epilogue)
        l = gm.GCNConv(dims, "relu", seed=5)
        a1, a2, b = both(gm, lambda: l(g, x))
        assert torch.equal(a1.view(torch.int32), b.view(torch.int32)) and torch.equal(a2.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("HC", [(8, 16), (1, 64), (4, 8), (2, 128)])
def test_gat_fold_is_bit_identical_to_the_combine_kernel(gm, HC):
    import torch
    H, C = HC
    rng = np.random.default_rng(H * 100 + C)
    n = 2500
    s, t = hub_graph(rng, n, 15000, HUBS)
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=n)
    x = torch.from_numpy(rng.standard_normal((n, 32), dtype=np.float32)).cuda()
    l = gm.GATConv((32, C), "relu", heads=H, seed=9)
    a1, a2, b = both(gm, lambda: l(g, x))
    assert torch.equal(a1.view(torch.int32), b.view(torch.int32)) and torch.equal(a2.view(torch.int32), b.view(torch.int32))


def test_fold_many_launches_and_streams(gm):
    """100 launches on one plan, alternating widths (the counters' layout depends on the tile count): every result equals the first"""
    import torch
    rng = np.random.default_rng(11)
    n = 1500
    s, t = hub_graph(rng, n, 8000, HUBS[:6])
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=n)
    xs = {D: torch.from_numpy(rng.standard_normal((n, D), dtype=np.float32)).cuda() for D in (8, 128, 520)}
    first = {D: gm.propagate(gm.copy_xj, g, "+", xj=x).clone() for D, x in xs.items()}
    for it in range(100):
        D = (8, 128, 520)[it % 3]
        y = gm.propagate(gm.copy_xj, g, "+", xj=xs[D])
        assert torch.equal(y, first[D]), it
