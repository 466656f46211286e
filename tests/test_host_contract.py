"""Host-layer contracts that round 1's review found unpinned: `degree`'s defaults and typing (GNNGraphs/src/query.jl:314-345),
index validation on every plan-building route (GNNGraphs/src/convert.jl:47-54), and the per-call handling of an `edge_weight`
argument in the k-hop layers (GNNlib/src/layers/conv.jl:501-542)."""
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


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_degree_defaults_match_the_reference(gm):
    """degree(g) on a DIRECTED graph is the out-degree, typed like the index vector; weighted graphs give Float32
    (query.jl:314-331: T = nothing, dir = :out, edge_weight = true)"""
    import torch
    s = np.array([1, 1, 1, 2, 4], np.int64)
    t = np.array([2, 3, 4, 3, 1], np.int64)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=4)
    d = gm.degree(g)
    assert d.dtype == torch.int64
    np.testing.assert_array_equal(d.cpu().numpy(), [3, 1, 0, 1])              # out-degrees, not in-degrees [1, 1, 2, 1]
    np.testing.assert_array_equal(gm.degree(g, dir="in").cpu().numpy(), [1, 1, 2, 1])
    assert gm.degree(g, torch.float32).dtype == torch.float32
    g32 = gm.GNNGraph(dev(s.astype(np.int32)), dev(t.astype(np.int32)), num_nodes=4)
    assert gm.degree(g32).dtype == torch.int32
    w = np.array([0.5, 0.25, 1.0, 2.0, 4.0], np.float32)
    gw = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=4)
    dw = gm.degree(gw)
    assert dw.dtype == torch.float32
    np.testing.assert_array_equal(dw.cpu().numpy(), [1.75, 2.0, 0.0, 4.0])
    assert gm.degree(gw, edge_weight=False).dtype == torch.int64              # counting edges: integer again (query.jl:303)


def test_every_plan_route_validates_unvalidated_indices(gm):
    """a graph marked _validated whose indices are NOT in range must still be refused by whichever plan is built first"""
    s = np.array([1, 2, 9], np.int64)       # 9 > num_nodes
    t = np.array([2, 3, 1], np.int64)
    for route in ("loops", "transposed", "degree_out", "plain"):
        g = gm.GNNGraph(dev(s), dev(t), num_nodes=3, _validated=True)
        g._indices_validated = False         # what a caller-built record looks like before any plan exists
        with pytest.raises(AssertionError):
            if route == "loops":
                g.plan(True)
            elif route == "transposed":
                g.plan_transposed()
            elif route == "degree_out":
                gm.degree(g)
            else:
                g.plan(False)
    # and a good graph is validated exactly once
    g = gm.GNNGraph(dev(np.array([1, 2], np.int64)), dev(np.array([2, 3], np.int64)), num_nodes=3)
    assert g._indices_validated


def test_batch_arrays_validates_each_member_against_its_own_size(gm):
    good = (np.array([1, 2], np.int64), np.array([2, 3], np.int64), 3)
    bad = (np.array([1, 4], np.int64), np.array([2, 1], np.int64), 3)       # 4 > 3: would alias node 1 of the next member
    with pytest.raises(AssertionError):
        gm.batch_arrays([bad, good])
    with pytest.raises(AssertionError):
        gm.batch_arrays([good, (np.array([0], np.int64), np.array([1], np.int64), 2)])
    g = gm.batch_arrays([good, good])
    assert g.num_nodes == 6 and g.num_edges == 4
    np.testing.assert_array_equal(g.s.cpu().numpy(), [1, 2, 4, 5])


def test_khop_edge_weight_argument_is_never_served_from_a_stale_cache(gm, oracle):
    """two different weight vectors that occupy the SAME device address (the first freed before the second is made) must give
    two different results — the failure mode of keying a cache on (data_ptr, _version)"""
    import torch
    from oracle import khop_layers as KO
    rng = np.random.default_rng(5)
    n, E, D = 300, 3000, 8
    s, t = rng.integers(1, n + 1, E), rng.integers(1, n + 1, E)
    x = rng.standard_normal((n, D)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gm.SGConv((D, D), k=2, seed=3)
    outs, ptrs = [], []
    for k in range(3):
        w = dev((rng.random(E) + 0.5).astype(np.float32) * (k + 1))
        ptrs.append(w.data_ptr())
        outs.append((l(g, dev(x), edge_weight=w).cpu().numpy(), w.cpu().numpy()))
        del w
        torch.cuda.synchronize()
    for y, w in outs:
        ref = KO.sg_conv(s, t, n, x, l.weight.cpu().numpy(), l.bias.cpu().numpy(), 2, edge_weight=w)
        assert np.linalg.norm(y - ref) <= 1e-5 * np.linalg.norm(ref)
    assert not any(k[0] == "gcn_norm_slots" and k[2] for k in g._cache if isinstance(k, tuple))   # nothing pinned


def test_segment_bounds_with_empty_graphs(gm, oracle):
    """reduce_nodes through the cached segment boundaries (gnnmp_segment_bounds + gnnmp_segment_pool_ptr_f32): graphs without
    nodes at the front, in the middle and at the end keep the identity of the aggregation, like NNlib.scatter"""
    rng = np.random.default_rng(9)
    gi = np.array([2, 2, 2, 4, 5, 5, 5, 5, 8, 8], np.int64)          # graphs 1, 3, 6, 7, 9, 10 are empty
    G, n, D = 10, len(gi), 12
    x = rng.standard_normal((n, D)).astype(np.float32)
    g = gm.GNNGraph(dev(np.array([1, 2], np.int64)), dev(np.array([2, 3], np.int64)), num_nodes=n, graph_indicator=dev(gi),
                    num_graphs=G)
    for aggr in ("+", "mean", "max", "min"):
        got = gm.reduce_nodes(aggr, g, dev(x)).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.scatter(aggr, x, gi, G))
    sp = g._cache["node_ptr"].cpu().numpy()
    np.testing.assert_array_equal(sp, [0, 0, 3, 3, 4, 8, 8, 8, 10, 10, 10])
