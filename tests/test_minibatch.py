"""induced_subgraph (GNNGraphs/src/sampling.jl:173-203) and NeighborLoader (samplers.jl:27-101) on the device.

induced_subgraph is index work: bit-exact against a line-by-line python restatement of the reference loop (and its own
docstring example / test, GNNGraphs/test/sampling.jl:49-66).  NeighborLoader is random: held to the invariants that
follow from its definition (every mini-batch is the induced subgraph of a node set that contains the batch's input
nodes, only nodes within num_layers hops, at most 1 + Σ_l Π K_l-bounded growth, batches partition the input nodes)."""
import numpy as np
import pytest


def ref_induced_subgraph(s, t, n, nodes):
    """sampling.jl:178-203 with neighbors(graph, node, dir = :in) = sources of the node's in-edges in edge order; the edge
    index recorded here is each edge's own position (the reference's findfirst returns the first parallel copy)"""
    node_map = {int(v): i + 1 for i, v in enumerate(nodes)}
    src, dst, eidx = [], [], []
    for v in nodes:
        for k in np.nonzero(t == v)[0]:
            if int(s[k]) in node_map:
                dst.append(node_map[int(v)])
                src.append(node_map[int(s[k])])
                eidx.append(k + 1)
    return np.array(src, dtype=s.dtype), np.array(dst, dtype=s.dtype), np.array(eidx, dtype=s.dtype)


def test_reference_example_on_the_restatement():
    # docstring example sampling.jl:139-170 and test/sampling.jl:49-60
    s, t = np.array([1, 2]), np.array([2, 3])
    a, b, e = ref_induced_subgraph(s, t, 3, np.array([1, 2, 3]))
    assert len(a) == 2 and (a == [1, 2]).all() and (b == [2, 3]).all()
    a, b, e = ref_induced_subgraph(s, t, 3, np.array([1, 2]))
    assert (a == [1]).all() and (b == [2]).all() and (e == [1]).all()


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("idx", ["int64", "int32"])
def test_induced_subgraph_bit_exact(gm, idx):
    from gnnmp import sampling as S
    rng = np.random.default_rng(12)
    n, E = 300, 5000
    s = rng.integers(1, n + 1, E).astype(idx)
    t = rng.integers(1, n + 1, E).astype(idx)
    x = rng.standard_normal((n, 7)).astype(np.float32)
    w = rng.random(E).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=n, x=dev(x))
    for nodes in (rng.permutation(n)[:60] + 1, np.arange(1, n + 1), np.array([5])):
        nodes = nodes.astype(idx)
        sub = S.induced_subgraph(g, dev(nodes))
        a, b, e = ref_induced_subgraph(s, t, n, nodes)
        assert sub.num_nodes == len(nodes) and sub.num_edges == len(a)
        np.testing.assert_array_equal(sub.s.cpu().numpy(), a)
        np.testing.assert_array_equal(sub.t.cpu().numpy(), b)
        np.testing.assert_array_equal(sub.eid.cpu().numpy(), e)
        np.testing.assert_array_equal(sub.nid.cpu().numpy(), nodes)
        np.testing.assert_array_equal(sub.x.cpu().numpy(), x[nodes - 1])           # getobs(graph.ndata, nodes)
        if len(e):
            np.testing.assert_array_equal(sub.w.cpu().numpy(), w[e - 1])
    # the reference's own example
    g2 = gm.GNNGraph(dev(np.array([1, 2])), dev(np.array([2, 3])), num_nodes=3)
    sub = S.induced_subgraph(g2, dev(np.array([1, 2, 3])))
    assert sub.num_nodes == 3 and sub.num_edges == 2
    sub = S.induced_subgraph(g2, dev(np.array([1, 2])))
    assert sub.num_nodes == 2 and sub.num_edges == 1


@pytest.mark.gpu
def test_node_set_is_ordered_and_deterministic(gm):
    from gnnmp import sampling as S
    g = gm.GNNGraph(dev(np.array([1, 2])), dev(np.array([2, 3])), num_nodes=50)
    ns = S.NodeSet(g)
    new = ns.add(dev(np.array([7, 3, 7, 9, 3, 3, 1])))
    assert new.cpu().tolist() == [7, 3, 9, 1]                 # each once, first occurrence first
    new = ns.add(dev(np.array([9, 4, 7, 4, 2])))
    assert new.cpu().tolist() == [4, 2]
    assert ns.nodes.cpu().tolist() == [7, 3, 9, 1, 4, 2]
    m = ns.map.cpu().numpy()
    assert [int(m[v - 1]) for v in (7, 3, 9, 1, 4, 2)] == [1, 2, 3, 4, 5, 6] and int((m != 0).sum()) == 6
    from gnnmp import _lib as L
    with pytest.raises(L.GnnmpError):
        ns.add(dev(np.array([51])))


def _hops(s, t, n, seeds, L):
    """nodes within L in-hops of the seeds"""
    reach = set(int(v) for v in seeds)
    frontier = set(reach)
    for _ in range(L):
        nxt = set()
        for v in frontier:
            nxt |= set(int(u) for u in s[t == v])
        frontier = nxt
        reach |= nxt
    return reach


@pytest.mark.gpu
def test_neighbor_loader_invariants(gm):
    from gnnmp import sampling as S
    rng = np.random.default_rng(21)
    n, E = 2000, 30000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    x = rng.standard_normal((n, 4)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n, x=dev(x))
    inputs = rng.permutation(n)[:23] + 1
    K = [3, 2]
    loader = S.NeighborLoader(g, num_neighbors=K, num_layers=2, input_nodes=dev(inputs), batch_size=5, seed=7)
    assert len(loader) == 5
    seen = []
    for b, mb in enumerate(loader):
        batch = inputs[b * 5:(b + 1) * 5]
        nid = mb.nid.cpu().numpy()
        seen += list(nid[:len(batch)])
        assert (nid[:len(batch)] == batch).all()                       # the batch's input nodes come first
        assert len(np.unique(nid)) == len(nid) == mb.num_nodes
        assert len(nid) <= len(batch) * (1 + K[0] + K[0] * K[1])       # growth bound of the sampling tree
        assert set(nid.tolist()) <= _hops(s, t, n, batch, 2)           # nothing beyond num_layers hops
        a, bb, e = ref_induced_subgraph(s, t, n, nid)                  # and the mini-batch IS the induced subgraph
        np.testing.assert_array_equal(mb.s.cpu().numpy(), a)
        np.testing.assert_array_equal(mb.t.cpu().numpy(), bb)
        np.testing.assert_array_equal(mb.x.cpu().numpy(), x[nid - 1])
    assert seen == list(inputs)                                        # the batches walk the input nodes in order
    # 0 neighbours per layer: the mini-batch is just the input nodes and the edges among them (test/samplers.jl:117-123)
    l0 = S.NeighborLoader(g, num_neighbors=[0], num_layers=1, input_nodes=dev(inputs[:4]), batch_size=2)
    for mb in l0:
        assert mb.num_nodes == 2
    # a larger batch size than input nodes: one batch (test/samplers.jl:101-113)
    l1 = S.NeighborLoader(g, num_neighbors=[2], num_layers=1, input_nodes=dev(inputs[:2]), batch_size=10)
    assert len(list(l1)) == 1
    # the same seed gives the same mini-batches
    a = [mb.nid.cpu().tolist() for mb in S.NeighborLoader(g, num_neighbors=K, num_layers=2, input_nodes=dev(inputs), batch_size=5, seed=7)]
    b = [mb.nid.cpu().tolist() for mb in loader]
    assert a == b
    # and a mini-batch feeds the path: one GCN layer on it runs on the HIP kernels
    mb = next(iter(loader))
    y = gm.GCNConv((4, 8), "relu", seed=1)(mb, mb.x)
    assert y.shape == (mb.num_nodes, 8)
