"""Graph-wise helpers of GNNlib/src/utils.jl (reduce_edges, softmax_nodes, softmax_edges, broadcast_nodes,
broadcast_edges).  CPU: the oracle restatement against the properties the reference's own tests assert
(GNNlib/test/utils.jl:22-56: batch of 5 graphs, 10 nodes / 60 directed edges each, Dx = 2, De = 3).  GPU: the HIP path
against the oracle."""
import numpy as np
import pytest


def ref_batch(seed=0, G=5, n=10, m=60, Dx=2, De=3):
    """MLUtils.batch of G rand_graph(n, m)-like members: block-diagonal edge index, sorted indicator"""
    rng = np.random.default_rng(seed)
    s, t, gi = [], [], []
    for k in range(G):
        a = rng.integers(0, n, m // 2)
        b = (a + 1 + rng.integers(0, n - 1, m // 2)) % n
        s.append(np.concatenate([a, b]) + 1 + k * n)
        t.append(np.concatenate([b, a]) + 1 + k * n)
        gi.append(np.full(n, k + 1))
    s, t, gi = np.concatenate(s), np.concatenate(t), np.concatenate(gi)
    x = rng.random((G * n, Dx)).astype(np.float32)
    e = rng.random((G * m, De)).astype(np.float32)
    return s, t, gi, x, e, G, n, m


def softmax64(a):
    a = a.astype(np.float64)
    p = np.exp(a - a.max(0, keepdims=True))
    return p / p.sum(0, keepdims=True)


@pytest.fixture(scope="module")
def GW(oracle):
    from oracle import graphwise
    return graphwise


def test_oracle_properties_of_reference_tests(oracle, GW):
    s, t, gi, x, e, G, n, m = ref_batch()
    r = GW.reduce_edges("mean", gi, s, e, G)                                  # test/utils.jl:22-26
    assert r.shape == (G, 3)
    np.testing.assert_allclose(r[1], e[m:2 * m].astype(np.float64).mean(0), rtol=1e-6)
    r = GW.softmax_nodes(gi, x, G)                                            # :28-32
    assert r.shape == x.shape
    np.testing.assert_allclose(r[:n], softmax64(x[:n]), rtol=1e-6)
    r = GW.softmax_edges(gi, s, e, G)                                         # :34-38
    assert r.shape == e.shape
    np.testing.assert_allclose(r[:m], softmax64(e[:m]), rtol=1e-6)
    z = np.random.default_rng(1).random((G, 4)).astype(np.float32)
    r = GW.broadcast_nodes(gi, z)                                             # :40-47
    assert r.shape == (G * n, 4)
    assert (r[0] == z[0]).all() and (r[n - 1] == z[0]).all() and (r[n] == z[1]).all()
    r = GW.broadcast_edges(gi, s, z)                                          # :49-56
    assert r.shape == (G * m, 4)
    assert (r[0] == z[0]).all() and (r[m - 1] == z[0]).all() and (r[m] == z[1]).all()


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
@pytest.mark.parametrize("G,n,m", [(5, 10, 60), (300, 33, 128), (2, 3000, 20000)])
def test_hip_graphwise_vs_oracle(gm, oracle, GW, idx, G, n, m):
    from gnnmp import utils as U
    s, t, gi, x, e, G, n, m = ref_batch(seed=G, G=G, n=n, m=m, Dx=5, De=3)
    g = gm.GNNGraph(dev(s.astype(idx)), dev(t.astype(idx)), num_nodes=G * n, graph_indicator=dev(gi.astype(idx)),
                    num_graphs=G)
    for aggr in ("+", "mean", "max", "min"):
        got = U.reduce_edges(aggr, g, dev(e)).cpu().numpy()
        ref = GW.reduce_edges(aggr, gi, s, e, G)
        if m <= 64:
            np.testing.assert_array_equal(got, ref)               # same order, same bits (segments not split)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    # softmax: exp is the device's, 1e-6-level
    got = U.softmax_nodes(g, dev(x)).cpu().numpy()
    assert np.abs(got - GW.softmax_nodes(gi, x, G)).max() <= 1e-6
    got = U.softmax_edges(g, dev(e)).cpu().numpy()
    assert np.abs(got - GW.softmax_edges(gi, s, e, G)).max() <= 1e-6
    z = np.random.default_rng(2).random((G, 4)).astype(np.float32)
    np.testing.assert_array_equal(U.broadcast_nodes(g, dev(z)).cpu().numpy(), GW.broadcast_nodes(gi, z))
    np.testing.assert_array_equal(U.broadcast_edges(g, dev(z)).cpu().numpy(), GW.broadcast_edges(gi, s, z))
    np.testing.assert_array_equal(gm.graph_indicator(g, edges=True).cpu().numpy(), GW.edge_indicator(gi, s))
