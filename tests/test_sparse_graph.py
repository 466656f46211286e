"""GNNGraph{SPARSE_T} at the device seam (GNNlib/ext/GNNlibAMDGPUExt.jl:13-32 dispatches on Union{COO_T, SPARSE_T}): the graph is a
compressed-sparse-column adjacency, edge_index(g) = findnz(A) (GNNGraphs/src/query.jl:14, convert.jl:62-73) is destination-sorted as
stored, and gnnmp_plan_from_csc takes the CSC arrays as the plan without a sort:
  * (rowptr, col, eid) bit-identical to gnnmp_plan_create on findnz(A)'s COO;
  * the reference's own known answers for graphs given as sparse matrices (GNNGraphs/test/gnngraph.jl:48-78, query.jl:49-87 run with
    GRAPH_T = :sparse, GNNGraphs/test/runtests.jl:48);
  * propagate / the layers on a sparse graph bit-identical to the same calls on the COO graph of findnz(A), and against the oracle."""
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


def random_csc(n, density, rng, hub=0):
    import scipy.sparse as sp
    A = sp.random(n, n, density=density, random_state=rng, format="csc", dtype=np.float32)
    A.data = (rng.standard_normal(A.nnz) * 0.7).astype(np.float32)
    if hub:
        A = A.tolil()
        A[rng.choice(n, hub, replace=False), 3] = 0.5         # column 3: a long row of the plan
        A = A.tocsc()
    A.sort_indices()
    return A


def findnz(A):
    """(s, t, v) 1-based in SparseMatrixCSC's findnz order: columns in order, rows ascending inside a column"""
    A = A.tocsc()
    t = np.repeat(np.arange(A.shape[1]), np.diff(A.indptr))
    return A.indices.astype(np.int64) + 1, t.astype(np.int64) + 1, A.data.astype(np.float32)


@pytest.mark.parametrize("n,density,hub,idx", [(1, 1.0, 0, np.int64), (50, 0.1, 0, np.int32), (3000, 0.004, 1500, np.int64), (20000, 0.0005, 0, np.int32)])
def test_plan_from_csc_is_plan_create_on_findnz(gm, n, density, hub, idx):
    import torch
    rng = np.random.default_rng(n)
    A = random_csc(n, density, rng, hub)
    s, t, v = findnz(A)
    colptr = torch.from_numpy(A.indptr.astype(idx) + 1).cuda()
    rowval = torch.from_numpy(A.indices.astype(idx) + 1).cuda()
    p = gm.Plan.from_csc(colptr, rowval, n, n)
    q = gm.Plan(torch.from_numpy(s.astype(idx)).cuda(), torch.from_numpy(t.astype(idx)).cuda(), n, n, 1, False)
    assert (p.n_dst, p.n_edges, p.n_total, p.max_degree, p.n_long) == (q.n_dst, q.n_edges, q.n_total, q.max_degree, q.n_long)
    if hub:
        assert p.n_long >= 1
    for name, a, b in zip(("rowptr", "col", "eid"), p.export(), q.export()):
        assert torch.equal(a, b), name
    assert torch.equal(p.export()[2].cpu(), torch.arange(A.nnz, dtype=torch.int32))       # slot k IS edge k
    s2, t2 = p.edge_index(torch.int64, 1)
    assert np.array_equal(s2.cpu().numpy(), s) and np.array_equal(t2.cpu().numpy(), t)


def test_zero_based_and_empty(gm):
    import torch
    colptr = torch.tensor([0, 0, 2, 2, 3], dtype=torch.int64).cuda()
    rowval = torch.tensor([0, 3, 1], dtype=torch.int64).cuda()
    p = gm.Plan.from_csc(colptr, rowval, 4, 4, index_base=0)
    rp, col, eid = (a.cpu().tolist() for a in p.export())
    assert rp == [0, 0, 2, 2, 3] and col == [0, 3, 1] and eid == [0, 1, 2]
    e = gm.Plan.from_csc(torch.ones(6, dtype=torch.int64).cuda(), torch.zeros(0, dtype=torch.int64).cuda(), 5, 5)
    assert (e.n_edges, e.max_degree) == (0, 0)
    g = gm.GNNGraph.from_sparse(colptr=torch.ones(6, dtype=torch.int64), rowval=torch.zeros(0, dtype=torch.int64))
    x = torch.randn((5, 3), device="cuda")
    assert torch.equal(gm.propagate(gm.copy_xj, g, "+", xj=x), torch.zeros_like(x))


@pytest.mark.parametrize("colptr,rowval", [([1, 3, 2, 4], [1, 2, 3]),         # decreasing
                                           ([2, 2, 3, 4], [1, 2, 3]),         # does not start at 1
                                           ([1, 2, 3, 3], [1, 2, 3]),         # does not end at E + 1
                                           ([1, 2, 3, 4], [1, 4, 3]),         # row index > n
                                           ([1, 2, 3, 4], [0, 1, 3])])        # row index < 1
def test_validation(gm, colptr, rowval):
    import torch
    for validate in (True, False):      # the structure is checked either way (a sanitised colptr [0,2,5,3,7] left row 1 unwritten)
        with pytest.raises(AssertionError):
            gm.Plan.from_csc(torch.tensor(colptr, dtype=torch.int64).cuda(), torch.tensor(rowval, dtype=torch.int64).cuda(), 3, 3,
                             validate=validate)


def test_stored_zeros_are_edges_like_findnz(gm):
    """num_edges of a sparse graph is nnz(A) (GNNGraphs/src/convert.jl:199) and edge_index is findnz(A) (convert.jl:62-73): an explicitly
    stored zero is an edge — it is counted, numbered, and contributes 0 * x_j to a weighted propagate and x_j to an unweighted one"""
    import scipy.sparse as sp
    import torch
    A = sp.csc_matrix((np.array([1.0, 0.0, 2.0], np.float32), np.array([1, 2, 0]), np.array([0, 2, 2, 3])), shape=(3, 3))
    assert A.nnz == 3                                     # scipy keeps the explicit zero at (2, 0)
    g = gm.GNNGraph.from_sparse(A)
    assert g.num_edges == 3
    s, t = g.plan().edge_index()
    assert (s.cpu().tolist(), t.cpu().tolist()) == ([2, 3, 1], [1, 1, 3])
    x = torch.tensor([[1.0], [10.0], [100.0]], device="cuda")
    assert gm.propagate(gm.copy_xj, g, "+", xj=x).cpu().flatten().tolist() == [110.0, 0.0, 1.0]
    assert gm.propagate(gm.w_mul_xj, g, "+", xj=x).cpu().flatten().tolist() == [10.0, 0.0, 2.0]
    assert gm.degree(g, dir="in", edge_weight=False).cpu().tolist() == [2, 0, 1]


def test_reference_known_answers_for_sparse_graphs(gm):
    """GNNGraphs/test/gnngraph.jl:48-78 and query.jl:49-87 with GRAPH_T = :sparse"""
    import scipy.sparse as sp
    import torch
    s = np.array([1, 1, 2, 2, 3, 3, 4, 4])
    t = np.array([2, 4, 1, 3, 2, 4, 1, 3])
    adj = np.array([[0, 1, 0, 1], [1, 0, 1, 0], [0, 1, 0, 1], [1, 0, 1, 0]])
    A = sp.csc_matrix((np.ones(8, np.float32), (s - 1, t - 1)), shape=(4, 4))
    g = gm.GNNGraph.from_sparse(A)
    assert gm.get_graph_type(g) == "sparse" and (g.num_nodes, g.num_edges) == (4, 8)
    s1, t1 = gm.sort_edge_index(*gm.edge_index(g))
    assert s1.cpu().tolist() == s.tolist() and t1.cpu().tolist() == t.tolist()                     # gnngraph.jl:73-75
    dense = np.zeros((4, 4), int)
    si, ti = (a.cpu().numpy() for a in gm.edge_index(g))
    dense[si - 1, ti - 1] = 1
    assert np.array_equal(dense, adj)                                                               # gnngraph.jl:81-84
    rp, col, _ = (a.cpu().numpy() for a in g.plan(False).export())
    assert sorted((col[rp[0]:rp[1]] + 1).tolist()) == [2, 4]                                        # inneighbors(g, 1), gnngraph.jl:71
    # degree, query.jl:49-56: sparse(s, t, 1) of a multigraph merges the two 1 -> 2 edges
    s, t = np.array([1, 1, 2, 3]), np.array([2, 2, 2, 4])
    A = sp.csc_matrix((np.ones(4, np.float32), (s - 1, t - 1)), shape=(4, 4))                       # duplicates summed, like sparse()
    g = gm.GNNGraph.from_sparse(A)
    assert g.num_edges == 3
    assert gm.degree(g).cpu().tolist() == [2, 1, 1, 0]                                              # query.jl:54-56: A[1, 2] = 2 counts twice
    assert gm.degree(g, dir="in").cpu().tolist() == [0, 3, 0, 1]
    assert gm.degree(g, dir="both").cpu().tolist() == [2, 4, 1, 1]
    assert gm.degree(g, edge_weight=False).cpu().tolist() == [1, 1, 1, 0]                           # query.jl:79-85 (not :coo)
    # weighted, query.jl:72-77: A[1, 2] = 0.1 + 2.1
    A = sp.csc_matrix((np.array([0.1, 2.1, 1.2, 1], np.float32), (s - 1, t - 1)), shape=(4, 4))
    g = gm.GNNGraph.from_sparse(A)
    np.testing.assert_allclose(gm.degree(g).cpu().numpy(), [2.2, 1.2, 1.0, 0.0], rtol=1e-6)


@pytest.mark.parametrize("aggr", ["+", "mean", "max", "min"])
def test_propagate_on_a_sparse_graph(gm, oracle, aggr):
    import torch
    rng = np.random.default_rng(5)
    n, D = 2000, 20
    A = random_csc(n, 0.004, rng)                 # (no split rows: their chunked sums are compared elsewhere with a tolerance)
    s, t, v = findnz(A)
    gs = gm.GNNGraph.from_sparse(A)
    gc = gm.GNNGraph(torch.from_numpy(s), torch.from_numpy(t), w=torch.from_numpy(v), num_nodes=n)
    x = rng.standard_normal((n, D), dtype=np.float32)
    e = rng.standard_normal((A.nnz, D), dtype=np.float32)
    xd, ed = torch.from_numpy(x).cuda(), torch.from_numpy(e).cuda()
    for f, kw, w in ((gm.copy_xj, dict(xj=xd), None), (gm.w_mul_xj, dict(xj=xd), v), (gm.e_mul_xj, dict(xj=xd, e=ed), False)):
        ys = gm.propagate(f, gs, aggr, **kw)
        yc = gm.propagate(f, gc, aggr, **kw)
        assert torch.equal(ys, yc), f.__name__
        if w is not False:                      # (e_mul_xj with a matrix e: against the COO graph's result only)
            ref = oracle.propagate(aggr, s, t, n, x, w=w)
            assert np.array_equal(ys.cpu().numpy(), ref, equal_nan=True), f.__name__


def test_layers_on_a_sparse_graph(gm, oracle):
    import torch
    rng = np.random.default_rng(9)
    n = 1500
    A = random_csc(n, 0.006, rng)
    A = (A + A.T).tocsc()
    A.data = np.abs(A.data) + 0.1                 # positive weights: the weighted GCN normalisation takes 1 / sqrt(degree)
    A.sort_indices()
    s, t, v = findnz(A)
    gs = gm.GNNGraph.from_sparse(A)
    gc = gm.GNNGraph(torch.from_numpy(s), torch.from_numpy(t), w=torch.from_numpy(v), num_nodes=n)
    x = torch.randn((n, 24), device="cuda")
    torch.manual_seed(0)
    for layer in (gm.GCNConv((24, 16), "relu", add_self_loops=False, seed=1), gm.SAGEConv((24, 16), "relu", seed=2), gm.GraphConv((24, 16), "relu", seed=3),
                  gm.GATConv((24, 8), heads=2, add_self_loops=False, seed=4)):
        assert torch.equal(layer(gs, x), layer(gc, x)), type(layer).__name__
    # the plan with self loops (GCNConv add_self_loops = true) needs the COO: materialised from the plan on demand
    gcn = gm.GCNConv((24, 16), "relu", add_self_loops=True, use_edge_weight=True, seed=5)
    y = gcn(gs, x)
    assert torch.isfinite(y).all() and torch.equal(y, gcn(gc, x))
    # ... and against the ORACLE on findnz(A)'s edge list (not only HIP against HIP): the four layer bodies of conv.jl on a sparse graph
    xh = x.cpu().numpy()

    def close(got, ref, what):
        got = got.cpu().numpy()
        assert np.linalg.norm(got - ref) <= 1e-5 * np.linalg.norm(ref), what
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), what
    l = gm.GCNConv((24, 16), "relu", add_self_loops=False, seed=1)
    close(l(gs, x), oracle.gcn_conv(s, t, n, xh, l.weight.cpu().numpy(), l.bias.cpu().numpy(), "relu", add_self_loops_=False), "GCNConv")
    close(y, oracle.gcn_conv(s, t, n, xh, gcn.weight.cpu().numpy(), gcn.bias.cpu().numpy(), "relu", add_self_loops_=True, use_edge_weight=True,
                             graph_w=v), "GCNConv(use_edge_weight)")
    l = gm.SAGEConv((24, 16), "relu", seed=2)
    close(l(gs, x), oracle.sage_conv(s, t, n, xh, l.weight.cpu().numpy(), l.bias.cpu().numpy(), "relu", aggr="mean", blas=False), "SAGEConv")
    l = gm.GraphConv((24, 16), "relu", seed=3)
    close(l(gs, x), oracle.graph_conv(s, t, n, xh, l.weight1.cpu().numpy(), l.weight2.cpu().numpy(), l.bias.cpu().numpy(), "relu", aggr="+",
                                      blas=False), "GraphConv")
    l = gm.GATConv((24, 8), heads=2, add_self_loops=False, seed=4)
    close(l(gs, x), oracle.gat_conv(s, t, n, xh, l.dense_x_weight.cpu().numpy(), l.a.cpu().numpy(), l.bias.cpu().numpy(), None, heads=2,
                                    add_self_loops_=False), "GATConv")
