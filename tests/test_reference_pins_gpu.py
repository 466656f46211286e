"""The reference's own known-answer / property tests for the hot path on the HIP PATH, through the C ABI — the twin of
tests/test_oracle_reference_pins.py (which pins the CPU oracle and is CPU-only): the same inputs and the same expected values, written
out, so that the pin travels with the driver's `-m gpu` run.  Each test cites the reference test it restates
(paths relative to /root/reference, which is NOT read here)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def dev(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def coo_from_adj(A):
    """findnz order (column-major), as GNNGraph(A; graph_type=:coo) builds it"""
    s, t = [], []
    n = A.shape[0]
    for j in range(n):
        for i in range(n):
            for _ in range(int(A[i, j])):
                s.append(i + 1)
                t.append(j + 1)
    return np.array(s, np.int64), np.array(t, np.int64)


def adjacency(s, t, n):
    A = np.zeros((n, n), np.int64)
    np.add.at(A, (np.asarray(s) - 1, np.asarray(t) - 1), 1)
    return A


# GraphNeuralNetworks/test/layers/conv.jl:30-44 — "edge weights & custom normalization"
def test_gcnconv_closed_form(gm):
    import torch
    s = np.array([2, 3, 1, 3, 1, 2])
    t = np.array([1, 1, 2, 2, 3, 3])
    w = np.array([1, 2, 3, 4, 5, 6], np.float32)
    g = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=3)
    x = torch.ones((3, 1), device="cuda")
    l = gm.GCNConv((1, 1), None, add_self_loops=False, use_edge_weight=True)
    l.weight = torch.ones((1, 1), device="cuda")
    d = gm.degree(g, dir="in").cpu().numpy()
    np.testing.assert_array_equal(d, [3, 7, 11])
    y = l(g, x).cpu().numpy()
    assert y[0, 0] == pytest.approx(w[0] / np.sqrt(d[0] * d[1]) + w[1] / np.sqrt(d[0] * d[2]), rel=RTOL)
    assert y[1, 0] == pytest.approx(w[2] / np.sqrt(d[1] * d[0]) + w[3] / np.sqrt(d[1] * d[2]), rel=RTOL)
    assert y[0, 0] == pytest.approx(0.5663732, rel=1e-6)
    assert y[1, 0] == pytest.approx(1.110496, rel=1e-6)
    # the edge_weight call argument gives the same result as the graph's own weights
    g0 = gm.GNNGraph(dev(s), dev(t), num_nodes=3)
    y2 = l(g0, x, edge_weight=dev(w)).cpu().numpy()
    np.testing.assert_allclose(y2, y, rtol=RTOL)


# GNNGraphs/test/query.jl:49-58, 73-87 — degree, unweighted and weighted
def test_degree(gm):
    s = np.array([1, 1, 2, 3])
    t = np.array([2, 2, 2, 4])
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=4)
    np.testing.assert_array_equal(gm.degree(g, dir="out").cpu().numpy(), [2, 1, 1, 0])
    np.testing.assert_array_equal(gm.degree(g, dir="in").cpu().numpy(), [0, 3, 0, 1])
    np.testing.assert_array_equal(gm.degree(g, dir="both").cpu().numpy(), [2, 4, 1, 1])
    w = np.array([0.1, 2.1, 1.2, 1], np.float32)
    gw = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=4)
    np.testing.assert_allclose(gm.degree(gw, dir="out").cpu().numpy(), [2.2, 1.2, 1.0, 0.0], rtol=1e-6)
    np.testing.assert_array_equal(gm.degree(gw, dir="out", edge_weight=False).cpu().numpy(), [2, 1, 1, 0])
    np.testing.assert_allclose(gm.degree(gw, dir="out", edge_weight=dev(2 * w)).cpu().numpy(), [4.4, 2.4, 2.0, 0.0], rtol=1e-6)


# GNNGraphs/test/transform.jl:1-17 — add self-loops (an existing loop becomes multiplicity 2)
def test_add_self_loops_adjacency(gm):
    A = np.array([[1, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [1, 0, 0, 0]])
    A2 = np.array([[2, 1, 0, 0], [0, 1, 1, 0], [0, 0, 1, 1], [1, 0, 0, 1]])
    s, t = coo_from_adj(A)
    g2 = gm.add_self_loops(gm.GNNGraph(dev(s), dev(t), num_nodes=4))
    s2, t2 = g2.s.cpu().numpy(), g2.t.cpu().numpy()
    np.testing.assert_array_equal(adjacency(s2, t2, 4), A2)
    assert len(s2) == A2.sum() and g2.w is None
    np.testing.assert_array_equal(s2[:len(s)], s)
    np.testing.assert_array_equal(s2[len(s):], [1, 2, 3, 4])
    np.testing.assert_array_equal(t2[len(s):], [1, 2, 3, 4])
    gw = gm.add_self_loops(gm.GNNGraph(dev(s), dev(t), dev(np.full(len(s), 0.5, np.float32)), num_nodes=4))
    np.testing.assert_array_equal(gw.w.cpu().numpy()[len(s):], np.ones(4, np.float32))       # transform.jl:22-24
    # the plan's fused self loops give the same propagate as the materialised ones
    import torch
    x = torch.rand((4, 3), device="cuda")
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=4)
    from gnnmp.msgpass import _fused
    from gnnmp import _lib as L
    assert torch.equal(_fused(g, L.COPY_XJ, "+", x, None, add_self_loops=True), gm.propagate(gm.copy_xj, g2, "+", xj=x))


# GNNGraphs/test/transform.jl:29-39 — batch
def test_batch_indicator_and_offsets(gm):
    rng = np.random.default_rng(0)

    def ring(n):
        p = rng.permutation(n)
        u, v = p, np.roll(p, 1)
        return np.concatenate([u, v]) + 1, np.concatenate([v, u]) + 1, n

    g1, g2, g3 = ring(10), ring(4), ring(7)
    for gb in (gm.batch_arrays([g1, g2, g3]), gm.batch([gm.GNNGraph(dev(a), dev(b), num_nodes=n) for a, b, n in (g1, g2, g3)])):
        np.testing.assert_array_equal(gb.graph_indicator.cpu().numpy(), [1] * 10 + [2] * 4 + [3] * 7)
        np.testing.assert_array_equal(gb.s.cpu().numpy(), np.concatenate([g1[0], 10 + g2[0], 14 + g3[0]]))
        np.testing.assert_array_equal(gb.t.cpu().numpy(), np.concatenate([g1[1], 10 + g2[1], 14 + g3[1]]))
        assert gb.num_nodes == 21 and gb.num_graphs == 3
    # ... and the same batch taken from a resident dataset (gnnmp_plan_select): getobs(g, 1:3) of the batched graph
    ds = gm.GraphDataset.from_members([g1, g2, g3])
    gs = ds.batch(np.arange(3))
    np.testing.assert_array_equal(gs.s.cpu().numpy(), np.concatenate([g1[0], 10 + g2[0], 14 + g3[0]]))
    np.testing.assert_array_equal(gs.graph_indicator.cpu().numpy(), [1] * 10 + [2] * 4 + [3] * 7)


def _test_graphs():
    """TEST_GRAPHS of GNNlib/test/test_module.jl:153-178: the 4-cycle and the graph with an isolated node"""
    adj1 = np.array([[0, 1, 0, 1], [1, 0, 1, 0], [0, 1, 0, 1], [1, 0, 1, 0]])
    adj2 = np.array([[0, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 1], [1, 0, 1, 0]])
    return [coo_from_adj(adj1) + (4,), coo_from_adj(adj2) + (4,)]


# GraphNeuralNetworks/test/layers/conv.jl:55-65 — conv_weight = zeros gives exactly zeros
def test_gcnconv_zero_conv_weight(gm):
    import torch
    l = gm.GCNConv((3, 5), None, seed=1)
    W0 = torch.zeros((5, 3), device="cuda")
    for s, t, n in _test_graphs():
        g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
        for x in (torch.ones((n, 3), device="cuda"), torch.rand((n, 3), device="cuda")):
            y = l(g, x, conv_weight=W0)
            assert y.shape == (n, 5) and float(y.abs().max()) == 0.0


# GNNlib/test/msgpass.jl:21-26 — isolated nodes
def test_propagate_isolated_nodes(gm):
    import torch
    x1 = torch.rand((6, 1), device="cuda")
    s = t = np.arange(1, 6)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=6)
    y1 = gm.propagate(gm.copy_xj, g, "+", xj=x1)
    assert y1.shape == (6, 1)
    assert torch.equal(y1[:5], x1[:5]) and float(y1[5, 0]) == 0.0


# GNNlib/test/msgpass.jl:69-116 — copy_xj / e_mul_xj / w_mul_xj  ≈  X * Adj   (n = 128, density 0.1, D = 10)
def test_propagate_matches_dense_matmul(gm):
    rng = np.random.default_rng(3)
    n = 128
    mask = rng.random((n, n)) < 0.1
    A = np.where(mask, rng.random((n, n)), 0.0)
    X = rng.random((n, 10)).astype(np.float32)
    s, t = coo_from_adj(mask.astype(np.int64))
    w = A[s - 1, t - 1].astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    gw = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=n)
    xd = dev(X)
    ref = mask.astype(np.float64).T @ X.astype(np.float64)
    np.testing.assert_allclose(gm.propagate(gm.copy_xj, g, "+", xj=xd).cpu().numpy(), ref, rtol=RTOL)
    refw = A.astype(np.float32).astype(np.float64).T @ X.astype(np.float64)
    np.testing.assert_allclose(gm.propagate(gm.e_mul_xj, g, "+", xj=xd, e=dev(w)).cpu().numpy(), refw, rtol=RTOL)
    np.testing.assert_allclose(gm.propagate(gm.w_mul_xj, gw, "+", xj=xd).cpu().numpy(), refw, rtol=RTOL)


# GNNlib/test/utils.jl:58-67 — softmax_edge_neighbors
def test_softmax_edge_neighbors(gm):
    s = np.array([1, 2, 3, 4])
    t = np.array([5, 5, 6, 6])
    e2 = np.random.default_rng(5).standard_normal((4, 3)).astype(np.float32)   # Julia (3, 4)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=6)
    z = gm.softmax_edge_neighbors(g, dev(e2)).cpu().numpy()

    def softmax(a):
        a = a.astype(np.float64)
        ex = np.exp(a - a.max(axis=0, keepdims=True))
        return ex / ex.sum(axis=0, keepdims=True)

    np.testing.assert_allclose(z[0:2], softmax(e2[0:2]), rtol=RTOL)
    np.testing.assert_allclose(z[2:4], softmax(e2[2:4]), rtol=RTOL)


# GNNlib/test/utils.jl:13-20 — reduce_nodes(mean) on a batch of 5 graphs; GraphNeuralNetworks/test/layers/pool.jl:4-20 — GlobalPool(+)
def test_reduce_nodes(gm):
    rng = np.random.default_rng(6)
    members = []
    for _ in range(5):
        p = rng.permutation(10)
        members.append((np.concatenate([p, np.roll(p, 1)]) + 1, np.concatenate([np.roll(p, 1), p]) + 1, 10))
    x = rng.random((50, 2), dtype=np.float32)
    g = gm.batch_arrays(members)
    r = gm.reduce_nodes("mean", g, dev(x)).cpu().numpy()
    assert r.shape == (5, 2)
    np.testing.assert_allclose(r[1], x[10:20].astype(np.float64).mean(axis=0), rtol=RTOL)
    u = gm.GlobalPool("+")(g, dev(x)).cpu().numpy()
    np.testing.assert_allclose(u[2], x[20:30].astype(np.float64).sum(axis=0), rtol=RTOL)
    one = gm.GlobalPool("+")(gm.GNNGraph(dev(members[0][0]), dev(members[0][1]), num_nodes=50), dev(x)).cpu().numpy()
    np.testing.assert_allclose(one[0], x.astype(np.float64).sum(axis=0), rtol=RTOL)      # single graph: indicator = ones (query.jl:500-505)


# layer shape contract on TEST_GRAPHS (GraphNeuralNetworks/test/layers/conv.jl:8-27,100-113,157-171,318-332)
def test_layer_shapes_on_test_graphs(gm):
    import torch
    D_IN, D_OUT = 3, 5
    for s, t, n in _test_graphs():
        g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
        x = torch.rand((n, D_IN), device="cuda")
        assert gm.GCNConv((D_IN, D_OUT), seed=1)(g, x).shape == (n, D_OUT)
        assert gm.GraphConv((D_IN, D_OUT), "relu", seed=2)(g, x).shape == (n, D_OUT)
        for aggr in ("mean", "max", "+"):
            assert gm.SAGEConv((D_IN, D_OUT), aggr=aggr, seed=3)(g, x).shape == (n, D_OUT)
        for heads in (1, 2):
            for concat in (True, False):
                y = gm.GATConv((D_IN, D_OUT), heads=heads, concat=concat, seed=4)(g, x)
                assert y.shape == (n, D_OUT * heads if concat else D_OUT) and bool(torch.isfinite(y).all())
