"""SGConv / TAGConv.  CPU: the oracle restatement against the dense float64 identity it must satisfy,
SGConv(k = 1) ≡ GCNConv without activation (the reference's own relationship: same normalisation, conv.jl:14-72 vs
:501-542).  GPU: the fused-hop HIP path against the oracle."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def KH(oracle):
    from oracle import khop_layers
    return khop_layers


def simple(rng, n, E):
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    k = s != t
    s, t = s[k], t[k]
    _, i = np.unique(s * 100000 + t, return_index=True)
    i = np.sort(i)
    return s[i], t[i]


def norm_adj(s, t, n, w=None, loops=True):
    A = np.zeros((n, n))
    np.add.at(A, (t - 1, s - 1), 1.0 if w is None else w.astype(np.float64))
    if loops:
        A += np.eye(n)
    d = A.sum(1)
    c = 1 / np.sqrt(d)
    return c[:, None] * A * c[None, :]


def rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("Din,Dout,k,weighted", [(6, 4, 1, False), (4, 6, 3, False), (5, 5, 2, True)])
def test_oracle_sg_and_tag_vs_dense_float64(oracle, KH, Din, Dout, k, weighted):
    rng = np.random.default_rng(Din * 10 + k)
    n = 40
    s, t = simple(rng, n, 300)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    W = (rng.standard_normal((Dout, Din)) * 0.4).astype(np.float32)
    b = (rng.standard_normal(Dout) * 0.1).astype(np.float32)
    w = (rng.random(len(s)) + 0.5).astype(np.float32) if weighted else None
    P = norm_adj(s, t, n, w)
    x64, W64 = x.astype(np.float64), W.astype(np.float64)
    y = KH.sg_conv(s, t, n, x, W, b, k=k, edge_weight=w)
    assert rel(y, np.linalg.matrix_power(P, k) @ x64 @ W64.T + b) < 5e-6
    y = KH.tag_conv(s, t, n, x, W, b, k=k, edge_weight=w)
    acc = np.zeros((n, Dout))
    sp = np.zeros((n, Din))
    for it in range(1, k + 1):
        sp = sp + np.linalg.matrix_power(P, it) @ x64
        acc = acc + sp @ W64.T
    assert rel(y, acc + b) < 5e-6


def test_oracle_sgconv_k1_is_gcnconv_identity(oracle, KH):
    rng = np.random.default_rng(3)
    n = 50
    s, t = simple(rng, n, 400)
    for Din, Dout in ((7, 3), (3, 7)):
        x = rng.standard_normal((n, Din)).astype(np.float32)
        W = (rng.standard_normal((Dout, Din)) * 0.4).astype(np.float32)
        b = (rng.standard_normal(Dout) * 0.1).astype(np.float32)
        np.testing.assert_array_equal(KH.sg_conv(s, t, n, x, W, b, k=1), oracle.gcn_conv(s, t, n, x, W, b, None))


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
@pytest.mark.parametrize("Din,Dout,k,weighted,loops", [(32, 16, 2, False, True), (16, 40, 3, False, True), (24, 24, 1, True, True),
                                                      (100, 100, 2, False, False), (8, 8, 3, True, False)])
def test_hip_sg_and_tag_vs_oracle(gm, KH, Din, Dout, k, weighted, loops):
    from gnnmp.layers_khop import SGConv, TAGConv
    rng = np.random.default_rng(Din + Dout + k)
    n, E = 1500, 20000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    t[:2500] = 4                                           # hub: split row
    if not loops:                                          # without self loops every node needs an in-edge (1/sqrt(0) = Inf)
        s = np.concatenate([s, np.arange(1, n + 1)])
        t = np.concatenate([t, np.roll(np.arange(1, n + 1), 1)])
    p = rng.permutation(len(s))
    s, t = s[p], t[p]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    w = (rng.random(len(s)) + 0.5).astype(np.float32) if weighted else None
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    for cls, fn in ((SGConv, KH.sg_conv), (TAGConv, KH.tag_conv)):
        l = cls((Din, Dout), k=k, add_self_loops=loops, seed=5)
        l.bias = dev((rng.standard_normal(Dout) * 0.1).astype(np.float32))
        y = l(g, dev(x), None if w is None else dev(w)).cpu().numpy()
        ref = fn(s, t, n, x, l.weight.cpu().numpy(), l.bias.cpu().numpy(), k=k, add_self_loops_=loops, edge_weight=w)
        assert y.shape == ref.shape
        assert rel(y, ref.astype(np.float64)) <= 1e-5
    # use_edge_weight = true takes the graph's own weights (conv.jl:524,531-532)
    if weighted:
        gw = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=n)
        l = SGConv((Din, Dout), k=k, add_self_loops=loops, use_edge_weight=True, seed=5)
        y = l(gw, dev(x)).cpu().numpy()
        ref = KH.sg_conv(s, t, n, x, l.weight.cpu().numpy(), l.bias.cpu().numpy(), k=k, add_self_loops_=loops, edge_weight=w)
        assert rel(y, ref.astype(np.float64)) <= 1e-5


def test_oracle_res_gated_vs_dense_float64(oracle, KH):
    rng = np.random.default_rng(8)
    n, Din, Dout = 40, 5, 6
    s, t = simple(rng, n, 300)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    A, B, U, V = ((rng.standard_normal((Dout, Din)) * 0.5).astype(np.float32) for _ in range(4))
    b = (rng.standard_normal(Dout) * 0.1).astype(np.float32)
    y = KH.res_gated_graph_conv(s, t, n, x, A, B, U, V, b, "relu")
    x64 = x.astype(np.float64)
    Ax, Bx, Ux, Vx = (x64 @ W.T.astype(np.float64) for W in (A, B, U, V))
    adj = np.zeros((n, n))
    adj[t - 1, s - 1] = 1
    gate = 1 / (1 + np.exp(-(Ax[:, None, :] + Bx[None, :, :])))           # [i, j, d]
    m = (adj[..., None] * gate * Vx[None]).sum(1)
    assert rel(y, np.maximum(Ux + m + b, 0)) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("Din,Dout", [(16, 32), (100, 100), (7, 5), (64, 1)])
def test_hip_res_gated_vs_oracle(gm, KH, Din, Dout):
    from gnnmp.layers_khop import ResGatedGraphConv
    rng = np.random.default_rng(Din * 3 + Dout)
    n, E = 1300, 18000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 5, E)                          # a few empty destinations
    t[:2000] = 6                                           # hub: split row
    p = rng.permutation(E)
    s, t = s[p], t[p]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = ResGatedGraphConv((Din, Dout), "relu", seed=9)
    l.bias = dev((rng.standard_normal(Dout) * 0.1).astype(np.float32))
    y = l(g, dev(x)).cpu().numpy()
    c = lambda v: v.cpu().numpy()
    ref = KH.res_gated_graph_conv(s, t, n, x, c(l.A), c(l.B), c(l.U), c(l.V), c(l.bias), "relu")
    assert y.shape == ref.shape
    assert rel(y, ref.astype(np.float64)) <= 1e-5
