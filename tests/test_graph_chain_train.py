"""The graph-classification chain's TRAINING step as one scheduled pullback (gnnmp.backward.graph_chain_ad, round 5): the model of
examples/graph_classification_tudataset.jl:79-82,97-104 — GNNChain(GraphConv(relu), GraphConv(relu), GlobalPool, Dense).  The fused entry
points (gnnmp_pool_grad_act_f32, gnnmp_dense_grad_w2_f32, gnnmp_propagate_add_mask_f32) do the arithmetic of the layer-by-layer adjoints
(graph_conv_ad / global_pool_ad / dense_ad: each held to 1e-5 of the oracle's rule-by-rule composition in tests/test_backward.py) with fewer
passes over the (N, D) arrays: logits and EVERY gradient must come out bit-identical to that composition; and against the oracle
directly (oracle.grad_graph_conv composed with the pooling / dense rules in float64) within 1e-5."""
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


def make(gm, G, dims, pool, aggr, seed, hub=False):
    import torch
    rng = np.random.default_rng(seed)
    members, xs = [], []
    for k in range(G):
        n = int(rng.integers(1, 41))
        m = int(rng.integers(0, 4 * n + 1))
        s, t = rng.integers(0, n, m), rng.integers(0, n, m)
        if hub and k % 7 == 0:
            s = np.concatenate([s, rng.integers(0, n, 150)]); t = np.concatenate([t, np.zeros(150, np.int64)])
        members.append((s.astype(np.int64) + 1, t.astype(np.int64) + 1, n))
        xs.append(rng.standard_normal((n, dims[0]), dtype=np.float32))
    g = gm.batch_arrays(members, xs)
    convs = [gm.GraphConv((dims[i], dims[i + 1]), "relu", aggr=aggr, seed=10 + i) for i in range(len(dims) - 1)]
    for c in convs:
        c.bias = torch.from_numpy(rng.standard_normal(c.bias.shape[0]).astype(np.float32) * 0.1).cuda()
    poolL, head = gm.GlobalPool(pool), gm.Dense((dims[-1], 2), seed=33)
    return g, members, xs, convs, poolL, head, rng


def params_of(convs, head):
    ps = []
    for c in convs:
        ps += [c.weight1, c.weight2, c.bias]
    return ps + [head.weight, head.bias]


@pytest.mark.parametrize("dims", [(16, 128, 128), (16, 64), (32, 48, 20, 128)])
@pytest.mark.parametrize("pool,aggr", [("mean", "+"), ("+", "+"), ("mean", "mean")])
def test_chain_pullback_is_bit_identical_to_the_layer_composition(gm, dims, pool, aggr):
    import torch
    from gnnmp.backward import dense_ad, global_pool_ad, graph_chain_ad, graph_conv_ad
    g, members, xs, convs, poolL, head, rng = make(gm, 300, dims, pool, aggr, 7 + len(dims), hub=True)
    ps = params_of(convs, head)
    r = torch.from_numpy(rng.standard_normal((g.num_graphs, 2)).astype(np.float32)).cuda()

    def run(fn, need_x):
        x = g.x.clone().requires_grad_(need_x)
        for p in ps:
            p.requires_grad_(True)
            p.grad = None
        y = fn(x)
        (y * r).sum().backward()
        out = [y.detach().clone()] + [p.grad.clone() for p in ps] + ([x.grad.clone()] if need_x else [])
        for p in ps:
            p.requires_grad_(False)
            p.grad = None
        return out

    def layered(x):
        h = x
        for c in convs:
            h = graph_conv_ad(c, g, h)
        return dense_ad(head, global_pool_ad(poolL, g, h))
    model = gm.GNNChain(*convs, poolL, head)
    for need_x in (False, True):
        a = run(lambda x: graph_chain_ad(model, g, x), need_x)
        b = run(layered, need_x)
        assert len(a) == len(b)
        for k, (u, v) in enumerate(zip(a, b)):
            assert u.shape == v.shape and torch.equal(u.view(torch.int32), v.view(torch.int32)), (k, float((u - v).abs().max()))


def test_chain_pullback_vs_oracle(gm, oracle):
    """the two-layer model of config 5 against the oracle's rule-by-rule composition (float64 dense rules, NNlib's scatter / gather rules)"""
    import torch
    from gnnmp.backward import graph_chain_ad
    dims = (16, 128, 128)
    g, members, xs, convs, poolL, head, rng = make(gm, 200, dims, "mean", "+", 5)
    ps = params_of(convs, head)
    r = rng.standard_normal((g.num_graphs, 2)).astype(np.float32)
    for p in ps:
        p.requires_grad_(True)
        p.grad = None
    y = graph_chain_ad(gm.GNNChain(*convs, poolL, head), g, g.x)
    (y * torch.from_numpy(r).cuda()).sum().backward()
    got = [p.grad.cpu().numpy() for p in ps]
    for p in ps:
        p.requires_grad_(False)
    # oracle: forward, then the rules backwards
    s, t, gi, n = oracle.batch(members)
    x = np.concatenate(xs)
    W = [p.detach().cpu().numpy() for p in ps]
    h1 = oracle.graph_conv(s, t, n, x, W[0], W[1], W[2], "relu", "+", blas=False)
    h2 = oracle.graph_conv(s, t, n, h1, W[3], W[4], W[5], "relu", "+", blas=False)
    pooled = oracle.global_pool("mean", gi, h2, len(members))
    ref_y = pooled.astype(np.float64) @ W[6].astype(np.float64).T + W[7][None, :]
    assert np.linalg.norm(y.detach().cpu().numpy() - ref_y) <= 1e-5 * np.linalg.norm(ref_y)
    dzh = r.astype(np.float64)
    dWh, dbh = dzh.T @ pooled.astype(np.float64), dzh.sum(0)
    dpool = dzh @ W[6].astype(np.float64)
    cnt = np.bincount(gi - 1, minlength=len(members)).astype(np.float64)
    dh2 = (dpool / cnt[:, None])[gi - 1].astype(np.float32)
    dh1, dW21, dW22, db2 = oracle.grad_graph_conv(s, t, n, h1, W[3], W[4], W[5], "relu", dh2, "+")
    _, dW11, dW12, db1 = oracle.grad_graph_conv(s, t, n, x, W[0], W[1], W[2], "relu", dh1, "+")
    ref = [dW11, dW12, db1, dW21, dW22, db2, dWh, dbh]
    for k, (a, b) in enumerate(zip(got, ref)):
        assert np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(b), (k, np.linalg.norm(a - b) / np.linalg.norm(b))
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), (k, np.abs(a - b).max() / np.abs(b).max())
