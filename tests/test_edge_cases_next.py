"""Empty / degenerate inputs through the entry points added after the headline path: no edges, isolated nodes, a single
node, one-feature rows, empty seed lists.  Each case is what the reference would return (checked against the oracle or a
closed form) or a clean GnnmpError — never a crash, a hang or garbage."""
import numpy as np
import pytest


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


E0 = np.zeros(0, np.int64)


@pytest.mark.gpu
def test_attention_layers_on_a_graph_without_edges(gm):
    """add_self_loops = true: every neighbourhood is the node itself, softmax weight 1 -> out_i = V_i"""
    import torch
    from gnnmp.layers_attn import AGNNConv, GATv2Conv, TransformerConv
    n = 37
    g = gm.GNNGraph(dev(E0), dev(E0), num_nodes=n)
    x = torch.randn((n, 12), device="cuda")
    l = GATv2Conv((12, 4), None, heads=2, bias=False, seed=1)
    y = l(g, x)
    Wxj = gm.dense(x, l.dense_j_weight)
    assert torch.allclose(y, Wxj, rtol=1e-6, atol=1e-7)
    assert torch.allclose(AGNNConv()(g, x), x, rtol=1e-6, atol=1e-7)
    t = TransformerConv((12, 4), heads=2, add_self_loops=True, root_weight=False, seed=2)
    assert torch.allclose(t(g, x), gm.dense(x, t.W2_weight, t.W2_bias), rtol=1e-6, atol=1e-7)
    # without self loops every neighbourhood is empty: the aggregate is 0 (NNlib scatter(+) identity)
    l2 = GATv2Conv((12, 4), None, heads=2, bias=False, add_self_loops=False, seed=1)
    assert bool((l2(g, x) == 0).all())
    assert bool((AGNNConv(add_self_loops=False)(g, x) == 0).all())


@pytest.mark.gpu
def test_gat_backward_on_isolated_nodes(gm, oracle):
    """a graph whose only edges are the added self loops: the pullback reduces to the dense rules"""
    import torch
    from gnnmp.backward import gat_conv_ad
    n = 50
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, 6)).astype(np.float32)
    r = rng.standard_normal((n, 8)).astype(np.float32)
    g = gm.GNNGraph(dev(E0), dev(E0), num_nodes=n)
    l = gm.GATConv((6, 4), None, heads=2, seed=3)
    for prm in (l.dense_x_weight, l.a, l.bias):
        prm.requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    (gat_conv_ad(l, g, xt) * dev(r)).sum().backward()
    dx, dW, da, db = oracle.grad_gat_conv(E0, E0, n, x, l.dense_x_weight.detach().cpu().numpy(), l.a.detach().cpu().numpy(),
                                          l.bias.detach().cpu().numpy(), None, r, heads=2)
    assert np.abs(xt.grad.cpu().numpy() - dx).max() <= 1e-5 * np.abs(dx).max()
    assert np.abs(l.dense_x_weight.grad.cpu().numpy() - dW).max() <= 1e-5 * np.abs(dW).max()
    assert np.abs(l.a.grad.cpu().numpy()).max() <= 1e-6            # α ≡ 1 does not depend on a
    assert np.abs(da).max() <= 1e-12


@pytest.mark.gpu
def test_message_functions_and_softmax_with_no_edges(gm):
    import torch
    n = 9
    g = gm.GNNGraph(dev(E0), dev(E0), num_nodes=n)
    x = torch.randn((n, 5), device="cuda")
    assert gm.apply_edges(gm.xi_dot_xj, g, xi=x, xj=x).shape == (0, 1)
    assert gm.apply_edges(gm.xi_sub_xj, g, xi=x, xj=x).shape == (0, 5)
    e = torch.empty((0, 5), device="cuda")
    y = gm.propagate(gm.e_mul_xj, g, "+", xj=x, e=e)
    assert y.shape == (n, 5) and bool((y == 0).all())
    y = gm.propagate(gm.e_mul_xj, g, "max", xj=x, e=e)
    assert bool(torch.isinf(y).all()) and bool((y < 0).all())       # NNlib: max over nothing = -Inf
    assert gm.softmax_edge_neighbors(g, torch.empty((0, 3), device="cuda")).shape == (0, 3)


@pytest.mark.gpu
def test_softmax_edge_neighbors_single_edge_rows_and_one_channel(gm, oracle):
    import torch
    s = np.array([1, 2, 3, 4, 4])
    t = np.array([5, 5, 6, 6, 7])                                  # GNNlib/test/utils.jl:58-66 shape + a 1-edge row
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=7)
    e = np.random.default_rng(1).standard_normal((5, 1)).astype(np.float32)
    z = gm.softmax_edge_neighbors(g, dev(e)).cpu().numpy()
    np.testing.assert_allclose(z, oracle.softmax_edge_neighbors(t, 7, e), rtol=1e-6)
    assert z[4, 0] == 1.0                                           # exp(0) / exp(0)
    np.testing.assert_allclose(z[0] + z[1], 1.0, rtol=1e-6)


@pytest.mark.gpu
def test_graph_prep_on_empty_and_tiny_inputs(gm):
    import torch
    from gnnmp import sampling as S
    u, v = S.sort_edge_index(dev(E0), dev(E0))
    assert u.numel() == 0 and v.numel() == 0
    g0 = gm.GNNGraph(dev(E0), dev(E0), num_nodes=4)
    assert S.is_bidirected(g0) is True and S.has_self_loops(g0) is False     # all(...) over nothing / any(...) over nothing
    sg = S.sample_neighbors(g0, dev(np.array([1, 2])), 3)
    assert sg.num_edges == 0 and sg.num_nodes == 4
    g = gm.GNNGraph(dev(np.array([1, 2, 3])), dev(np.array([2, 3, 1])), num_nodes=3)
    sg = S.sample_neighbors(g, dev(E0), 2)                                    # no seeds
    assert sg.num_edges == 0
    sub = S.induced_subgraph(g, dev(np.array([2])))                           # one node, no edge inside
    assert sub.num_nodes == 1 and sub.num_edges == 0
    sub = S.induced_subgraph(g, dev(np.array([3, 1, 2])))                     # every node, permuted
    assert sub.num_edges == 3
    # relabelled: node 3 -> 1, 1 -> 2, 2 -> 3; edges grouped by target in list order: (2->3)=(3->1'), (3->1)=(1'->2'), (1->2)=(2'->3')
    assert sub.s.cpu().tolist() == [3, 1, 2] and sub.t.cpu().tolist() == [1, 2, 3]
    loader = S.NeighborLoader(g, num_neighbors=[2], num_layers=1, input_nodes=dev(np.array([1])))
    mbs = list(loader)
    assert len(mbs) == 1 and set(mbs[0].nid.cpu().tolist()) == {1, 3}        # node 1 and its only in-neighbour


@pytest.mark.gpu
def test_row_partition_with_more_ranks_than_rows_with_edges(gm):
    import torch
    from gnnmp import rowpart as RP
    t = torch.tensor([2, 2, 2, 2], dtype=torch.int64)
    b = RP.partition_rows_by_edges(t, 5, 4)
    assert b[0][0] == 0 and b[-1][1] == 5 and all(lo <= hi for lo, hi in b)
    assert sum(int(((t - 1 >= lo) & (t - 1 < hi)).sum()) for lo, hi in b) == 4


@pytest.mark.gpu
def test_more_layers_on_degenerate_graphs(gm):
    """CGConv / EdgeConv / GatedGraphConv / DConv with no edges, a single node, a single edge"""
    import torch
    from oracle import more_layers as ML
    rng = np.random.default_rng(2)
    for n, s, t in ((9, E0, E0), (1, E0, E0), (1, np.array([1]), np.array([1])), (4, np.array([2]), np.array([3]))):
        g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
        x = rng.standard_normal((n, 6)).astype(np.float32)
        E = len(s)
        # CGConv with edge features
        e = rng.standard_normal((E, 3)).astype(np.float32)
        l = gm.CGConv(((6, 3), 6), "softplus", residual=True, seed=1)
        y = l(g, dev(x), dev(e) if E else torch.empty((0, 3), device="cuda")).cpu().numpy()
        ref = ML.cg_conv(s, t, n, x, e, l.dense_f_weight.cpu().numpy(), l.dense_f_bias.cpu().numpy(),
                         l.dense_s_weight.cpu().numpy(), l.dense_s_bias.cpu().numpy(), "softplus", True)
        np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6)
        # EdgeConv: empty destinations keep the identity of the aggregation (-Inf for max, 0 for +)
        d1 = gm.Dense((12, 5), "relu", seed=2)
        nn = [(d1.weight.cpu().numpy(), d1.bias.cpu().numpy(), "relu")]
        for aggr in ("max", "+"):
            y = gm.EdgeConv(d1, aggr=aggr)(g, dev(x)).cpu().numpy()
            ref = ML.edge_conv(s, t, n, x, nn, aggr)
            assert y.shape == ref.shape == (n, 5)
            np.testing.assert_array_equal(np.isfinite(y), np.isfinite(ref))
            fin = np.isfinite(ref)
            np.testing.assert_allclose(y[fin], ref[fin], rtol=1e-5, atol=1e-6)
        # GatedGraphConv (input padded from 6 to 8 features), DConv
        lg = gm.GatedGraphConv(8, 2, seed=3)
        y = lg(g, dev(x)).cpu().numpy()
        ref = ML.gated_graph_conv(s, t, n, x, lg.weight.cpu().numpy(), lg.gru_Wi.cpu().numpy(), lg.gru_Wh.cpu().numpy(),
                                  lg.gru_b.cpu().numpy(), "+")
        np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6)
        ld = gm.DConv((6, 4), 3, seed=4)
        y = ld(g, dev(x)).cpu().numpy()
        ref = ML.d_conv(s, t, n, x, ld.weights.cpu().numpy(), ld.bias.cpu().numpy(), 3)
        np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_agnn_adjoint_without_edges_and_layer_argument_checks(gm):
    import torch
    from gnnmp.backward_attn import agnn_conv_ad
    from gnnmp.layers_attn import AGNNConv
    n = 11
    g = gm.GNNGraph(dev(E0), dev(E0), num_nodes=n)
    x = torch.randn((n, 5), device="cuda", requires_grad=True)
    l = AGNNConv()
    l.beta = torch.tensor([1.5], device="cuda", requires_grad=True)
    y = agnn_conv_ad(l, g, x)                              # only self loops: out = x, dL/dx = r, dL/dβ = 0
    r = torch.randn((n, 5), device="cuda")
    (y * r).sum().backward()
    assert torch.allclose(y, x, rtol=1e-6, atol=1e-7)
    assert torch.allclose(x.grad, r, rtol=1e-5, atol=1e-6)
    assert abs(float(l.beta.grad[0])) < 1e-5
    with pytest.raises(AssertionError):
        gm.GatedGraphConv(4, 1)(g, torch.randn((n, 5), device="cuda"))      # more input than output features
    with pytest.raises(AssertionError):
        gm.CGConv((5, 5))(g, torch.randn((n + 1, 5), device="cuda"))        # wrong number of nodes
    with pytest.raises(AssertionError):
        gm.CGConv(((5, 2), 5))(g, torch.randn((n, 5), device="cuda"))       # built with edge features, none given
