"""GATConv with edge features (l.dense_e; GNNlib/src/layers/conv.jl:112-167 with e !== nothing, add_self_loops = false).
CPU: the oracle restatement against a float64 edge-list formulation.  GPU: the one-pass kernel with the per-edge logit
term against the oracle; the reference's own assertions (conv.jl:114-121) are raised by the mirror."""
import numpy as np
import pytest


def _problem(seed, n, E, Din, ein, H, C):
    rng = np.random.default_rng(seed)
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 3, E)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    e = rng.standard_normal((E, ein)).astype(np.float32)
    W = (rng.standard_normal((H * C, Din)) / np.sqrt(Din)).astype(np.float32)
    We = (rng.standard_normal((H * C, ein)) / np.sqrt(ein)).astype(np.float32)
    a = (rng.standard_normal((3 * C, H)) * 0.6).astype(np.float32)
    b = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    return s, t, x, e, W, We, a, b


def test_oracle_gat_edge_features_vs_float64(oracle):
    from oracle import attn_layers as AL
    n, E, H, C = 40, 300, 2, 3
    s, t, x, e, W, We, a, b = _problem(1, n, E, 5, 4, H, C)
    y = AL.gat_conv_edge(s, t, n, x, e, W, We, a, b, "relu", heads=H)
    Wx = (x.astype(np.float64) @ W.T).reshape(n, H, C)
    Wee = (e.astype(np.float64) @ We.T).reshape(E, H, C)
    a64 = a.astype(np.float64).T                                       # [H, 3C]
    z = (np.concatenate([Wx[t - 1], Wx[s - 1], Wee], 2) * a64[None]).sum(2)
    l = np.where(z > 0, z, 0.2 * z)
    out = np.zeros((n, H, C))
    for i in range(n):
        k = np.nonzero(t - 1 == i)[0]
        if len(k):
            p = np.exp(l[k] - l[k].max(0))
            al = p / p.sum(0)
            out[i] = (al[..., None] * Wx[s[k] - 1]).sum(0)
    ref = np.maximum(out.reshape(n, H * C) + b, 0)
    assert np.linalg.norm(y - ref) <= 5e-6 * np.linalg.norm(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din,ein,concat", [(8, 16, 100, 12, True), (4, 8, 20, 3, False), (1, 7, 9, 5, True), (2, 4, 6, 1, True)])
def test_hip_gat_edge_features_vs_oracle(oracle, H, C, Din, ein, concat):
    import torch
    import gnnmp
    from oracle import attn_layers as AL
    gnnmp.load()
    n, E = 1300, 21000
    s, t, x, e, W, We, a, b = _problem(H * 10 + C, n, E, Din, ein, H, C)
    t[:2600] = 9                                   # hub: split row
    p = np.random.default_rng(0).permutation(E)
    s, t, e = s[p], t[p], e[p]
    if not concat:
        b = b[:C]
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    g = gnnmp.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gnnmp.GATConv(((Din, ein), C), "relu", heads=H, concat=concat, add_self_loops=False, seed=1)
    l.dense_x_weight, l.dense_e_weight, l.a, l.bias = dev(W), dev(We), dev(a), dev(b)
    y = l(g, dev(x), dev(e)).cpu().numpy()
    ref = AL.gat_conv_edge(s, t, n, x, e, W, We, a, b, "relu", heads=H, concat=concat)
    assert y.shape == ref.shape
    assert np.linalg.norm(y - ref) <= 1e-5 * np.linalg.norm(ref)
    # the reference's assertions (conv.jl:114-121)
    with pytest.raises(AssertionError):
        l(g, dev(x))                                                   # edge features required for this layer
    with pytest.raises(AssertionError):
        gnnmp.GATConv((Din, C), heads=H, add_self_loops=False, seed=1)(g, dev(x), dev(e))    # not declared in the constructor
    with pytest.raises(AssertionError):
        gnnmp.GATConv(((Din, ein), C), heads=H, add_self_loops=True)   # edge features + self loops
    with pytest.raises(AssertionError):
        l(g, dev(x), dev(e[:-1]))                                      # wrong number of edge features
