"""Pullbacks of SGConv / TAGConv.  Both layers are LINEAR in x and in the weight, so the HIP gradients are checked against
the oracle FORWARD through the adjoint identities  <J_x x', r> = <x', J_xᵀ r>  and  <J_W W', r> = <W', J_Wᵀ r>  (J applied by
the oracle with the bias removed), plus Δb = column sums of r."""
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


@pytest.mark.gpu
@pytest.mark.parametrize("layer,Din,Dout,k,weighted,loops", [("sg", 12, 8, 2, False, True), ("sg", 8, 12, 3, True, True),
                                                            ("tag", 10, 10, 3, False, True), ("tag", 6, 9, 2, True, False)])
def test_hip_khop_backward_adjoint_identity(gm, oracle, layer, Din, Dout, k, weighted, loops):
    from oracle import khop_layers as KH
    from gnnmp.backward_khop import sg_conv_ad, tag_conv_ad
    rng = np.random.default_rng(Din * 7 + k)
    n, E = 1300, 15000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    t[:2000] = 7
    s[3000:4500] = 11
    ring = np.arange(1, n + 1)
    s, t = np.concatenate([s, ring]), np.concatenate([t, np.roll(ring, 1)])       # every node has an in-edge
    w = (rng.random(len(s)) + 0.5).astype(np.float32) if weighted else None
    x = rng.standard_normal((n, Din)).astype(np.float32)
    W = (rng.standard_normal((Dout, Din)) * 0.4).astype(np.float32)
    b = (rng.standard_normal(Dout) * 0.1).astype(np.float32)
    r = rng.standard_normal((n, Dout)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), None if w is None else dev(w), num_nodes=n)
    cls, fn, ofn = (gm.SGConv, sg_conv_ad, KH.sg_conv) if layer == "sg" else (gm.TAGConv, tag_conv_ad, KH.tag_conv)
    l = cls((Din, Dout), k, add_self_loops=loops, use_edge_weight=weighted)
    l.weight, l.bias = dev(W).requires_grad_(True), dev(b).requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    y = fn(l, g, xt)
    ref = ofn(s, t, n, x, W, b, k=k, add_self_loops_=loops, edge_weight=w)
    assert np.linalg.norm(y.detach().cpu().numpy() - ref) <= 1e-5 * np.linalg.norm(ref)
    (y * dev(r)).sum().backward()
    r64 = r.astype(np.float64)
    for _ in range(3):
        xp = rng.standard_normal((n, Din)).astype(np.float32)
        lhs = (ofn(s, t, n, xp, W, None, k=k, add_self_loops_=loops, edge_weight=w).astype(np.float64) * r64).sum()
        rhs = (xp.astype(np.float64) * xt.grad.cpu().numpy()).sum()
        assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), np.linalg.norm(xp) * np.linalg.norm(xt.grad.cpu().numpy()) * 1e-2)
        Wp = rng.standard_normal((Dout, Din)).astype(np.float32)
        lhs = (ofn(s, t, n, x, Wp, None, k=k, add_self_loops_=loops, edge_weight=w).astype(np.float64) * r64).sum()
        rhs = (Wp.astype(np.float64) * l.weight.grad.cpu().numpy()).sum()
        assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), np.linalg.norm(Wp) * np.linalg.norm(l.weight.grad.cpu().numpy()) * 1e-2)
    np.testing.assert_allclose(l.bias.grad.cpu().numpy(), r64.sum(0), rtol=2e-5, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("aggr,D,Dout", [("+", 16, 12), ("mean", 10, 10)])
def test_hip_gin_backward_vs_rule_by_rule_float64(gm, aggr, D, Dout):
    """gin_conv with nn = Dense(relu): y = relu(W z + b), z = (1 + ϵ) x + aggr_j x_j — every pullback rule written out"""
    from gnnmp.backward_khop import gin_conv_ad
    rng = np.random.default_rng(D)
    n, E, eps = 1200, 14000, 0.3
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    t[:1500] = 5
    x = rng.standard_normal((n, D)).astype(np.float32)
    r = rng.standard_normal((n, Dout)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    nn = gm.Dense((D, Dout), "relu", seed=4)
    nn.bias = dev((rng.standard_normal(Dout) * 0.1).astype(np.float32))
    l = gm.GINConv(nn, eps, aggr=aggr)
    nn.weight.requires_grad_(True)
    nn.bias.requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    y = gin_conv_ad(l, g, xt)
    assert bool((y == l(g, dev(x))).all())                          # same forward as the plain layer
    (y * dev(r)).sum().backward()
    W, b = nn.weight.detach().cpu().numpy().astype(np.float64), nn.bias.detach().cpu().numpy().astype(np.float64)
    x64 = x.astype(np.float64)
    A = np.zeros((n, n))
    np.add.at(A, (t - 1, s - 1), 1.0)
    if aggr == "mean":
        A = A / np.maximum(A.sum(1, keepdims=True), 1)
    z = (1 + np.float32(eps)).astype(np.float64) * x64 + A @ x64
    pre = z @ W.T + b
    dz = (r * (pre > 0)) @ W
    want = {"x": (1 + np.float32(eps)).astype(np.float64) * dz + A.T @ dz, "W": (r * (pre > 0)).T @ z, "b": (r * (pre > 0)).sum(0)}
    safe = np.abs(pre) > 1e-4                                        # relu kinks: fp32 / fp64 may disagree on the sign there
    assert safe.mean() > 0.999
    for got, ref in ((xt.grad, want["x"]), (nn.weight.grad, want["W"]), (nn.bias.grad, want["b"])):
        gn = got.cpu().numpy()
        assert np.linalg.norm(gn - ref) <= 2e-4 * np.linalg.norm(ref)
