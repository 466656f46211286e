"""GATv2Conv / AGNNConv / TransformerConv / GINConv.

CPU: the oracle's statement-by-statement restatement (oracle/attn_layers.py, float32, per-edge temporaries like the
reference) against an independent float64 dense-adjacency formulation of the same layer — guards the restatement, since
the reference holds no known-answer vectors for these layers.
GPU: the one-pass HIP kernel (gnnmp_attn_conv_f32) against the oracle, 1e-5 relative (north_star's fp32 bar)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def AL(oracle):
    from oracle import attn_layers
    return attn_layers


def simple_graph(rng, n, E):
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    keep = s != t
    s, t = s[keep], t[keep]
    _, k = np.unique(s * 100000 + t, return_index=True)
    k = np.sort(k)
    return s[k], t[k]


def rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)


# ------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("H,C,concat,loops", [(2, 4, True, True), (3, 5, False, True), (1, 7, True, False)])
def test_oracle_gatv2_vs_dense_float64(oracle, AL, H, C, concat, loops):
    rng = np.random.default_rng(H * 10 + C)
    n, Din = 50, 6
    s, t = simple_graph(rng, n, 400)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    Wi = (rng.standard_normal((H * C, Din)) * 0.5).astype(np.float32)
    bi = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    Wj = (rng.standard_normal((H * C, Din)) * 0.5).astype(np.float32)
    a = rng.standard_normal((C, H)).astype(np.float32)
    b = (rng.standard_normal(H * C if concat else C) * 0.1).astype(np.float32)
    y = AL.gatv2_conv(s, t, n, x, Wi, bi, Wj, a, b, "relu", heads=H, concat=concat, add_self_loops_=loops)
    Q = (x.astype(np.float64) @ Wi.T.astype(np.float64) + bi).reshape(n, H, C)
    K = (x.astype(np.float64) @ Wj.T.astype(np.float64)).reshape(n, H, C)

    def logits():
        z = Q[:, None] + K[None]                                  # [i, j, H, C]
        return (np.where(z > 0, z, 0.2 * z) * a.T.astype(np.float64)[None, None]).sum(-1)

    o = AL.dense_attention_f64(s, t, n, logits, K, H, C, loops)
    o = o.reshape(n, H * C) if concat else o.mean(axis=1)
    ref = np.maximum(o + b, 0)
    if not loops:
        has = np.bincount(t - 1, minlength=n) > 0
        assert np.all(y[~has] == np.maximum(b, 0))               # empty neighbourhood: zero aggregate
    assert rel(y, ref) < 5e-6


@pytest.mark.parametrize("loops", [True, False])
def test_oracle_agnn_vs_dense_float64(oracle, AL, loops):
    rng = np.random.default_rng(5)
    n, D = 60, 9
    s, t = simple_graph(rng, n, 500)
    x = rng.standard_normal((n, D)).astype(np.float32)
    y = AL.agnn_conv(s, t, n, x, beta=1.7, add_self_loops_=loops)
    xn = x.astype(np.float64) / np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True)
    o = AL.dense_attention_f64(s, t, n, lambda: (1.7 * xn @ xn.T)[..., None], x, 1, D, loops).reshape(n, D)
    assert rel(y, o) < 5e-6


@pytest.mark.parametrize("H,C,concat,root,skip", [(2, 4, True, True, False), (2, 3, False, True, False), (1, 6, True, False, True)])
def test_oracle_transformer_vs_dense_float64(oracle, AL, H, C, concat, root, skip):
    rng = np.random.default_rng(H + C)
    n = 40
    Din = H * C if skip else 5
    s, t = simple_graph(rng, n, 300)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    mk = lambda r: ((rng.standard_normal((r, Din)) * 0.5).astype(np.float32), (rng.standard_normal(r) * 0.1).astype(np.float32))
    om = H * C if concat else C
    (W1, b1), (W2, b2), (W3, b3), (W4, b4) = mk(om), mk(H * C), mk(H * C), mk(H * C)
    if not root:
        W1 = b1 = None
    y = AL.transformer_conv(s, t, n, x, W1, b1, W2, b2, W3, b3, W4, b4, heads=H, concat=concat, add_self_loops_=True,
                            skip_connection=skip)
    x64 = x.astype(np.float64)
    lin = lambda W, b: x64 @ W.T.astype(np.float64) + b
    Q, K, V = lin(W3, b3).reshape(n, H, C), lin(W4, b4).reshape(n, H, C), lin(W2, b2)
    o = AL.dense_attention_f64(s, t, n, lambda: np.einsum("ihc,jhc->ijh", Q, K) / np.sqrt(C), V, H, C, True)
    o = o.reshape(n, H * C) if concat else o.mean(axis=1)
    if root:
        o = o + lin(W1, b1)
    if skip:
        o = o + x64
    assert rel(y, o) < 5e-6


def test_oracle_gin_pre_nn(oracle, AL):
    rng = np.random.default_rng(9)
    n, D = 30, 4
    s, t = simple_graph(rng, n, 200)
    x = rng.standard_normal((n, D)).astype(np.float32)
    z = AL.gin_conv(s, t, n, x, 0.25)
    A = np.zeros((n, n))
    A[t - 1, s - 1] = 1
    assert rel(z, 1.25 * x.astype(np.float64) + A @ x.astype(np.float64)) < 2e-6


# ------------------------------------------------------------------------------------------------- GPU
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


def hub_graph(rng, n, E):
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    t[: E // 8] = 5                    # hub destination: a split row (adaptive threshold 64 at this size)
    p = rng.permutation(E)
    return s[p], t[p]


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,concat,Din", [(8, 16, True, 100), (4, 8, False, 20), (1, 64, True, 32), (2, 4, True, 6),
                                           (1, 25, True, 10), (1, 3, False, 4), (16, 4, True, 12),
                                           (4, 7, False, 10), (2, 12, True, 9), (5, 3, True, 7)])   # odd head widths
def test_hip_gatv2_vs_oracle(gm, AL, H, C, concat, Din):
    from gnnmp.layers_attn import GATv2Conv
    rng = np.random.default_rng(H * 100 + C)
    n, E = 1200, 20000
    s, t = hub_graph(rng, n, E)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    l = GATv2Conv((Din, C), "relu", heads=H, concat=concat, seed=7)
    l.dense_i_bias = dev((rng.standard_normal(H * C) * 0.1).astype(np.float32))
    l.bias = dev((rng.standard_normal(H * C if concat else C) * 0.1).astype(np.float32))
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    y = l(g, dev(x)).cpu().numpy()
    ref = AL.gatv2_conv(s, t, n, x, l.dense_i_weight.cpu().numpy(), l.dense_i_bias.cpu().numpy(),
                        l.dense_j_weight.cpu().numpy(), l.a.cpu().numpy(), l.bias.cpu().numpy(), "relu", heads=H,
                        concat=concat)
    assert y.shape == ref.shape
    assert rel(y, ref.astype(np.float64)) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("D,loops", [(128, True), (100, True), (7, False), (1, True), (256, True), (64, False)])
def test_hip_agnn_vs_oracle(gm, AL, D, loops):
    from gnnmp.layers_attn import AGNNConv
    rng = np.random.default_rng(D)
    n, E = 1000, 16000
    s, t = hub_graph(rng, n, E)
    x = rng.standard_normal((n, D)).astype(np.float32)
    l = AGNNConv(init_beta=1.3, add_self_loops=loops)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    y = l(g, dev(x)).cpu().numpy()
    ref = AL.agnn_conv(s, t, n, x, beta=1.3, add_self_loops_=loops)
    assert rel(y, ref.astype(np.float64)) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,concat,root,skip,loops", [(8, 16, True, True, False, False), (4, 8, False, True, False, True),
                                                       (1, 32, True, False, True, True), (2, 64, True, True, False, False)])
def test_hip_transformer_vs_oracle(gm, AL, H, C, concat, root, skip, loops):
    from gnnmp.layers_attn import TransformerConv
    rng = np.random.default_rng(H * 7 + C)
    n, E = 1100, 18000
    s, t = hub_graph(rng, n, E)
    Din = H * C if skip else 24
    x = rng.standard_normal((n, Din)).astype(np.float32)
    l = TransformerConv((Din, C), heads=H, concat=concat, add_self_loops=loops, root_weight=root, skip_connection=skip, seed=3)
    for nm in ("W1_bias", "W2_bias", "W3_bias", "W4_bias"):
        b = getattr(l, nm)
        if b is not None:
            setattr(l, nm, dev((rng.standard_normal(b.numel()) * 0.1).astype(np.float32)))
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    y = l(g, dev(x)).cpu().numpy()
    c = lambda v: None if v is None else v.cpu().numpy()
    ref = AL.transformer_conv(s, t, n, x, c(l.W1_weight), c(l.W1_bias), c(l.W2_weight), c(l.W2_bias), c(l.W3_weight),
                              c(l.W3_bias), c(l.W4_weight), c(l.W4_bias), heads=H, concat=concat, add_self_loops_=loops,
                              skip_connection=skip)
    assert y.shape == ref.shape
    assert rel(y, ref.astype(np.float64)) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("aggr", ["+", "mean", "max"])
def test_hip_gin_vs_oracle(gm, oracle, AL, aggr):
    from gnnmp.layers_attn import GINConv
    rng = np.random.default_rng(2)
    n, E, D = 900, 12000, 20
    s, t = hub_graph(rng, n, E)
    s = np.concatenate([s, np.arange(1, n + 1)])
    t = np.concatenate([t, np.roll(np.arange(1, n + 1), 1)])       # every node receives something (max of nothing = -Inf)
    x = rng.standard_normal((n, D)).astype(np.float32)
    nn = gm.Dense((D, 12), "relu", seed=1)
    l = GINConv(nn, 0.3, aggr=aggr)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    y = l(g, dev(x)).cpu().numpy()
    z = AL.gin_conv(s, t, n, x, 0.3, aggr=aggr)
    ref = oracle._act("relu", oracle.matmul(nn.weight.cpu().numpy(), z) + nn.bias.cpu().numpy()[None, :])
    assert rel(y, ref.astype(np.float64)) <= 1e-5


@pytest.mark.gpu
def test_attn_conv_rejects_what_it_cannot_do(gm):
    import torch
    from gnnmp import _lib as L
    from gnnmp.layers_attn import ATTN_COS, ATTN_GATV2, attn_conv
    g = gm.GNNGraph(dev(np.array([1, 2, 3])), dev(np.array([2, 3, 1])), num_nodes=3)
    plan = g.plan(True)
    K = torch.randn((3, 2 * 130), device="cuda")
    a = torch.randn((2, 130), device="cuda")
    with pytest.raises(L.GnnmpError) as ei:                    # 260 features = 130 float2 lanes: wider than a wave
        attn_conv(plan, ATTN_GATV2, K, a=a, H=2, C=130)
    assert ei.value.status == L.EUNSUPPORTED
    with pytest.raises(L.GnnmpError) as ei:                    # cosine logit is single-head
        attn_conv(plan, ATTN_COS, K[:, :24].contiguous(), H=2, C=12)
    assert ei.value.status == L.EINVAL
