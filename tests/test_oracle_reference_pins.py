"""Pins the CPU oracle against every known-answer / property test the reference holds for the hot path
(SURVEY.md §8c).  Each test cites the reference test it re-states (paths relative to /root/reference, which is NOT read
here: inputs and expected values are written out).  CPU only.
"""
import numpy as np
import pytest

RTOL = 1e-5


def adjacency(s, t, n, w=None):
    A = np.zeros((n, n), np.float64)
    for k in range(len(s)):
        A[s[k] - 1, t[k] - 1] += 1.0 if w is None else w[k]
    return A


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


# GraphNeuralNetworks/test/layers/conv.jl:30-44 — "edge weights & custom normalization"
def test_gcnconv_closed_form(oracle):
    s = np.array([2, 3, 1, 3, 1, 2])
    t = np.array([1, 1, 2, 2, 3, 3])
    w = np.array([1, 2, 3, 4, 5, 6], np.float32)
    x = np.ones((3, 1), np.float32)
    W = np.ones((1, 1), np.float32)
    b = np.zeros(1, np.float32)
    d = oracle.degree(t, 3, w)
    np.testing.assert_array_equal(d, [3, 7, 11])
    y = oracle.gcn_conv(s, t, 3, x, W, b, add_self_loops_=False, use_edge_weight=True, graph_w=w)
    e1 = w[0] / np.sqrt(d[0] * d[1]) + w[1] / np.sqrt(d[0] * d[2])
    e2 = w[2] / np.sqrt(d[1] * d[0]) + w[3] / np.sqrt(d[1] * d[2])
    assert y[0, 0] == pytest.approx(e1, rel=RTOL)
    assert y[1, 0] == pytest.approx(e2, rel=RTOL)
    assert y[0, 0] == pytest.approx(0.5663732, rel=1e-6)
    assert y[1, 0] == pytest.approx(1.110496, rel=1e-6)
    # `y ≈ l(g, x, w; norm_fn = custom_norm_fn)` — the edge_weight call argument gives the same result
    y2 = oracle.gcn_conv(s, t, 3, x, W, b, add_self_loops_=False, use_edge_weight=True, graph_w=w, edge_weight=w)
    np.testing.assert_allclose(y2, y, rtol=RTOL)
    # the CPU SpMM fast path agrees (reference asserts only ≈ between the two paths)
    y3 = oracle.gcn_conv(s, t, 3, x, W, b, add_self_loops_=False, use_edge_weight=True, graph_w=w, fast_path=True)
    np.testing.assert_allclose(y3, y, rtol=RTOL)


# GNNGraphs/test/query.jl:49-58 — degree, unweighted
def test_degree_unweighted(oracle):
    s = np.array([1, 1, 2, 3])
    t = np.array([2, 2, 2, 4])
    np.testing.assert_array_equal(oracle.degree(s, 4), [2, 1, 1, 0])            # dir = :out
    np.testing.assert_array_equal(oracle.degree(t, 4), [0, 3, 0, 1])            # dir = :in
    np.testing.assert_array_equal(oracle.degree(s, 4) + oracle.degree(t, 4), [2, 4, 1, 1])  # :both
    assert oracle.degree(s, 4).dtype == np.float32


# GNNGraphs/test/query.jl:73-87 — degree, weighted
def test_degree_weighted(oracle):
    s = np.array([1, 1, 2, 3])
    w = np.array([0.1, 2.1, 1.2, 1], np.float32)
    np.testing.assert_allclose(oracle.degree(s, 4, w), [2.2, 1.2, 1.0, 0.0], rtol=1e-6)
    np.testing.assert_array_equal(oracle.degree(s, 4, None), [2, 1, 1, 0])       # edge_weight = false
    np.testing.assert_allclose(oracle.degree(s, 4, 2 * w), [4.4, 2.4, 2.0, 0.0], rtol=1e-6)


# GNNGraphs/test/transform.jl:1-17 — add self-loops (an existing loop becomes multiplicity 2)
def test_add_self_loops_adjacency(oracle):
    A = np.array([[1, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [1, 0, 0, 0]])
    A2 = np.array([[2, 1, 0, 0], [0, 1, 1, 0], [0, 0, 1, 1], [1, 0, 0, 1]])
    s, t = coo_from_adj(A)
    assert len(s) == A.sum()
    s2, t2, w2 = oracle.add_self_loops(s, t, 4)
    assert w2 is None
    np.testing.assert_array_equal(adjacency(s2, t2, 4), A2)
    assert len(s2) == A2.sum()
    # appended after the existing edges, in node order
    np.testing.assert_array_equal(s2[:len(s)], s)
    np.testing.assert_array_equal(s2[len(s):], [1, 2, 3, 4])
    np.testing.assert_array_equal(t2[len(s):], [1, 2, 3, 4])
    # weighted graphs get weight-1 loops (transform.jl:22-24)
    _, _, w3 = oracle.add_self_loops(s, t, 4, np.full(len(s), 0.5, np.float32))
    np.testing.assert_array_equal(w3[len(s):], np.ones(4, np.float32))


# GNNGraphs/test/transform.jl:29-39 — batch
def test_batch_indicator_and_offsets(oracle):
    rng = np.random.default_rng(0)

    def ring(n):  # random_regular_graph(n, 2): any 2-regular graph will do for the index arithmetic
        p = rng.permutation(n)
        u, v = p, np.roll(p, 1)
        return np.concatenate([u, v]) + 1, np.concatenate([v, u]) + 1, n

    g1, g2, g3 = ring(10), ring(4), ring(7)
    s, t, gi, n = oracle.batch([g1, g2, g3])
    np.testing.assert_array_equal(gi, [1] * 10 + [2] * 4 + [3] * 7)
    np.testing.assert_array_equal(s, np.concatenate([g1[0], 10 + g2[0], 14 + g3[0]]))
    np.testing.assert_array_equal(t, np.concatenate([g1[1], 10 + g2[1], 14 + g3[1]]))
    assert n == 21


def _test_graphs():
    """TEST_GRAPHS of GNNlib/test/test_module.jl:153-178: the 4-cycle and the graph with an isolated node"""
    adj1 = np.array([[0, 1, 0, 1], [1, 0, 1, 0], [0, 1, 0, 1], [1, 0, 1, 0]])
    adj2 = np.array([[0, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 1], [1, 0, 1, 0]])
    return [coo_from_adj(adj1) + (4,), coo_from_adj(adj2) + (4,)]


# GraphNeuralNetworks/test/layers/conv.jl:55-65 — conv_weight = zeros gives exactly zeros
def test_gcnconv_zero_conv_weight(oracle):
    D_IN, D_OUT = 3, 5
    W0 = np.zeros((D_OUT, D_IN), np.float32)
    b = np.zeros(D_OUT, np.float32)
    rng = np.random.default_rng(1)
    for s, t, n in _test_graphs():
        for x in (np.ones((n, D_IN), np.float32), rng.random((n, D_IN), dtype=np.float32)):
            y = oracle.gcn_conv(s, t, n, x, W0, b)
            assert y.shape == (n, D_OUT)
            np.testing.assert_array_equal(y, np.zeros((n, D_OUT), np.float32))


# GNNlib/test/msgpass.jl:21-26 — isolated nodes
def test_propagate_isolated_nodes(oracle):
    x1 = np.random.default_rng(2).random((6, 1), dtype=np.float32)
    s = t = np.arange(1, 6)
    y1 = oracle.propagate("+", s, t, 6, x1)
    assert y1.shape == (6, 1)
    np.testing.assert_array_equal(y1[:5], x1[:5])
    assert y1[5, 0] == 0.0


# GNNlib/test/msgpass.jl:69-89 — copy_xj fused and unfused ≈ X * Adj   (n = 128, density 0.1, D = 10)
def test_propagate_copy_xj_matches_dense_matmul(oracle):
    rng = np.random.default_rng(3)
    n = 128
    Adj = (rng.random((n, n)) < 0.1).astype(np.float64)
    X = rng.random((n, 10)).astype(np.float32)
    s, t = coo_from_adj(Adj)
    ref = (Adj.T @ X.astype(np.float64))          # Julia: X (10, n) * Adj -> column i = sum_j Adj[j, i] X[:, j]
    unfused = oracle.propagate("+", s, t, n, X)
    fused = oracle.spmm_csc(s, t, n, X)
    np.testing.assert_allclose(unfused, ref, rtol=RTOL)
    np.testing.assert_allclose(fused, ref, rtol=RTOL)


# GNNlib/test/msgpass.jl:91-116 — e_mul_xj / w_mul_xj ≈ X * A
def test_propagate_weighted_matches_dense_matmul(oracle):
    rng = np.random.default_rng(4)
    n = 128
    mask = rng.random((n, n)) < 0.1
    A = np.where(mask, rng.random((n, n)), 0.0)
    X = rng.random((n, 10)).astype(np.float32)
    s, t = coo_from_adj(mask.astype(np.int64))
    w = A[s - 1, t - 1].astype(np.float32)
    ref = A.astype(np.float32).astype(np.float64).T @ X.astype(np.float64)
    np.testing.assert_allclose(oracle.propagate("+", s, t, n, X, w), ref, rtol=RTOL)
    np.testing.assert_allclose(oracle.spmm_csc(s, t, n, X, w), ref, rtol=RTOL)


# GNNlib/test/utils.jl:58-67 — softmax_edge_neighbors
def test_softmax_edge_neighbors(oracle):
    s = np.array([1, 2, 3, 4])
    t = np.array([5, 5, 6, 6])
    e2 = np.random.default_rng(5).standard_normal((4, 3)).astype(np.float32)   # Julia (3, 4)
    z = oracle.softmax_edge_neighbors(t, 6, e2)
    assert z.shape == e2.shape

    def softmax(a):  # over the edge dimension
        a = a.astype(np.float64)
        ex = np.exp(a - a.max(axis=0, keepdims=True))
        return ex / ex.sum(axis=0, keepdims=True)

    np.testing.assert_allclose(z[0:2], softmax(e2[0:2]), rtol=RTOL)
    np.testing.assert_allclose(z[2:4], softmax(e2[2:4]), rtol=RTOL)


# GNNlib/test/utils.jl:13-20 — reduce_nodes(mean) on a batch of 5 graphs;
# GraphNeuralNetworks/test/layers/pool.jl:4-20 — GlobalPool(+)
def test_reduce_nodes(oracle):
    rng = np.random.default_rng(6)
    ns = [10] * 5
    gi = np.concatenate([np.full(n, i + 1) for i, n in enumerate(ns)])
    x = rng.random((50, 2), dtype=np.float32)
    r = oracle.reduce_nodes("mean", gi, x)
    assert r.shape == (5, 2)
    np.testing.assert_allclose(r[1], x[10:20].astype(np.float64).mean(axis=0), rtol=RTOL)
    u = oracle.global_pool("+", gi, x)
    np.testing.assert_allclose(u[2], x[20:30].astype(np.float64).sum(axis=0), rtol=RTOL)
    # single graph: graph_indicator = ones (GNNGraphs/src/query.jl:500-505)
    one = oracle.global_pool("+", np.ones(50, np.int64), x)
    np.testing.assert_allclose(one[0], x.astype(np.float64).sum(axis=0), rtol=RTOL)


# layer shape contract on TEST_GRAPHS (GraphNeuralNetworks/test/layers/conv.jl:8-27,100-113,157-171,318-332)
def test_layer_shapes_on_test_graphs(oracle):
    D_IN, D_OUT = 3, 5
    rng = np.random.default_rng(7)
    for s, t, n in _test_graphs():
        x = rng.random((n, D_IN), dtype=np.float32)
        W = rng.standard_normal((D_OUT, D_IN)).astype(np.float32)
        b = np.zeros(D_OUT, np.float32)
        assert oracle.gcn_conv(s, t, n, x, W, b).shape == (n, D_OUT)
        assert oracle.graph_conv(s, t, n, x, W, W, b, "relu", "+").shape == (n, D_OUT)
        for aggr in ("mean", "max", "+"):
            W2 = rng.standard_normal((D_OUT, 2 * D_IN)).astype(np.float32)
            y = oracle.sage_conv(s, t, n, x, W2, b, None, aggr)
            assert y.shape == (n, D_OUT)
        for heads in (1, 2):
            for concat in (True, False):
                Wd = rng.standard_normal((D_OUT * heads, D_IN)).astype(np.float32)
                a = rng.standard_normal((2 * D_OUT, heads)).astype(np.float32)
                bb = np.zeros(D_OUT * heads if concat else D_OUT, np.float32)
                y = oracle.gat_conv(s, t, n, x, Wd, a, bb, None, heads, concat)
                assert y.shape == (n, D_OUT * heads if concat else D_OUT)
                assert np.isfinite(y).all()


# --- dual-restatement cross-checks (C loop vs independent numpy twin) ------------------------------------------
@pytest.mark.parametrize("aggr", ["+", "mean", "max", "min"])
@pytest.mark.parametrize("D", [1, 3, 16])
def test_c_oracle_equals_numpy_twin_bitwise(oracle, oracle_np, aggr, D):
    rng = np.random.default_rng(10 + D)
    n, E = 37, 400                       # multi-edges and self loops are certain, some rows stay empty
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 4, E)        # nodes n-5..n have no incoming edge
    x = rng.standard_normal((n, D)).astype(np.float32)
    w = rng.random(E).astype(np.float32)
    a = oracle.propagate(aggr, s, t, n, x)
    b = oracle_np.propagate(aggr, s, t, n, x)
    np.testing.assert_array_equal(a, b)
    a = oracle.propagate(aggr, s, t, n, x, w)
    b = oracle_np.propagate(aggr, s, t, n, x, w)
    np.testing.assert_array_equal(a, b)
    # empty destinations keep the identity (NNlib fill): 0 for + and mean, -Inf for max, +Inf for min
    ident = {"+": 0.0, "mean": 0.0, "max": -np.inf, "min": np.inf}[aggr]
    assert (a[n - 5:] == ident).all()


def test_empty_graph(oracle):
    x = np.ones((5, 4), np.float32)
    e = np.zeros(0, np.int64)
    np.testing.assert_array_equal(oracle.propagate("+", e, e, 5, x), np.zeros((5, 4), np.float32))
    assert (oracle.propagate("max", e, e, 5, x) == -np.inf).all()
    np.testing.assert_array_equal(oracle.spmm_csc(e, e, 5, x), np.zeros((5, 4), np.float32))


def test_spmm_matches_scipy(oracle, oracle_np):
    rng = np.random.default_rng(11)
    n, E = 200, 3000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    x = rng.standard_normal((n, 7)).astype(np.float32)
    w = rng.random(E).astype(np.float32)
    for ww in (None, w):
        ref = oracle_np.spmm_scipy(s, t, n, x, ww)
        scale = np.abs(ref).max()
        assert np.abs(oracle.spmm_csc(s, t, n, x, ww) - ref).max() <= 1e-5 * scale
        assert np.abs(oracle.propagate("+", s, t, n, x, ww) - ref).max() <= 1e-5 * scale


def test_softmax_twin(oracle, oracle_np):
    rng = np.random.default_rng(12)
    n, E, H = 30, 300, 4
    t = rng.integers(1, n + 1, E)
    e = (3 * rng.standard_normal((E, H))).astype(np.float32)
    a = oracle.softmax_edge_neighbors(t, n, e)
    b = oracle_np.softmax_edge_neighbors(t, n, e)
    np.testing.assert_allclose(a, b, rtol=2e-6)
    sums = oracle_np.scatter("+", a, t, n)
    has = np.bincount(t - 1, minlength=n) > 0
    np.testing.assert_allclose(sums[has], 1.0, rtol=1e-5)


def test_index_out_of_range_is_reported(oracle):
    with pytest.raises(IndexError):
        oracle.propagate("+", np.array([1, 9]), np.array([1, 2]), 3, np.ones((3, 2), np.float32))


def test_reference_microbenchmark_isequal_pins_the_summation_order(oracle):
    """GraphNeuralNetworks/perf/bench_gnn.jl:7-40 holds an EXACT assertion, not an approximate one:
        A = sprand(n, n, 0.01); B = rand(100, n); g = GNNGraph(A; graph_type = :coo)
        @assert isequal(propagate((xi, xj, e) -> e .* xj, g, +; xj = B, e = A.nzval'), B * A)      # and the same for e_mul_xj
    i.e. the reference's generic gather -> message -> scatter(+) result is BIT-EQUAL to the dense x CSC product, whose loop order is
    fixed by SparseArrays (for col, for k in nzrange(col): C[:, col] += B[:, row_k] * val_k).  Both sides agree only if scatter(+) adds
    the messages of a destination in edge order (= findnz order = ascending row inside a column) starting from zero, with the product
    rounded before the add — which is how the oracle restates NNlib's CPU scatter.  One of the few places where the reference itself
    pins the summation ORDER (otherwise 'parity unpinned'); checked in Float64 (the benchmark's eltype) and Float32."""
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    n = 256
    A = sp.random(n, n, density=0.03, format="csc", random_state=rng, dtype=np.float64)
    A.sort_indices()
    t = np.repeat(np.arange(1, n + 1), np.diff(A.indptr)).astype(np.int64)      # findnz(A): column by column, rows ascending
    s = (A.indices + 1).astype(np.int64)
    for dt in (np.float64, np.float32):
        B = rng.random((n, 20)).astype(dt)
        e = A.data.astype(dt)
        ref = np.zeros((n, 20), dt)
        for col in range(n):
            for k in range(A.indptr[col], A.indptr[col + 1]):
                ref[col] = ref[col] + B[A.indices[k]] * e[k]
        got = oracle.propagate("+", s, t, n, B, e)
        assert got.dtype == dt and np.array_equal(got, ref), dt
        if dt == np.float32:
            assert np.array_equal(oracle.spmm_csc(s, t, n, B, e), ref)          # the fast path's restatement is that loop too
