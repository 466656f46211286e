"""GATConv with concat = false (heads averaged before bias and σ, GNNlib/src/layers/conv.jl:143-147): the pullback of
`mean(x, dims = 2)` in front of the attention pullback.  CPU: oracle adjoint vs finite differences; GPU: HIP vs oracle."""
import numpy as np
import pytest


def _problem(seed, n, E, Din, H, C):
    rng = np.random.default_rng(seed)
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    k = s != t
    s, t = s[k], t[k]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    W = (rng.standard_normal((H * C, Din)) / np.sqrt(Din)).astype(np.float32)
    a = (rng.standard_normal((2 * C, H)) * 0.7).astype(np.float32)
    b = (rng.standard_normal(C) * 0.1).astype(np.float32)
    r = rng.standard_normal((n, C)).astype(np.float32)
    return s, t, x, W, a, b, r


def test_oracle_gat_mean_heads_adjoint_vs_finite_differences(oracle):
    n, H, C = 40, 3, 4
    s, t, x, W, a, b, r = _problem(5, n, 260, 6, H, C)

    def loss(xv, Wv, av, bv):
        y = oracle.gat_conv(s, t, n, xv, Wv, av, bv, "relu", heads=H, concat=False)
        return float((y.astype(np.float64) * r).sum())

    grads = oracle.grad_gat_conv(s, t, n, x, W, a, b, "relu", r, heads=H, concat=False)
    args = [x, W, a, b]
    rng = np.random.default_rng(3)
    eps, checked = 2e-3, 0
    for which, grad in enumerate(grads):
        for _ in range(10):
            idx = tuple(int(rng.integers(0, d)) for d in args[which].shape)
            ap = [v.copy() for v in args]
            am = [v.copy() for v in args]
            ap[which][idx] += eps
            am[which][idx] -= eps
            f0, fp, fm = loss(*args), loss(*ap), loss(*am)
            if abs((fp - f0) - (f0 - fm)) > 0.05 * eps * max(1.0, abs(float(grad[idx]))):
                continue
            assert (fp - fm) / (2 * eps) == pytest.approx(float(grad[idx]), rel=3e-2, abs=3e-2)
            checked += 1
    assert checked >= 25


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din", [(4, 8, 20), (8, 16, 100), (2, 4, 6)])
def test_hip_gat_mean_heads_backward_vs_oracle(oracle, H, C, Din):
    import torch
    import gnnmp
    from gnnmp.backward import gat_conv_ad
    gnnmp.load()
    n, E = 1500, 24000
    s, t, x, W, a, b, r = _problem(H * 10 + C, n, E, Din, H, C)
    s[:2500] = 11
    t[3000:6000] = 7
    keep = s != t
    s, t = s[keep], t[keep]
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    g = gnnmp.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gnnmp.GATConv((Din, C), "relu", heads=H, concat=False, seed=3)
    l.dense_x_weight, l.a, l.bias = dev(W), dev(a), dev(b)
    for p in (l.dense_x_weight, l.a, l.bias):
        p.requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    y = gat_conv_ad(l, g, xt)
    ref = oracle.gat_conv(s, t, n, x, W, a, b, "relu", heads=H, concat=False)
    assert y.shape == ref.shape == (n, C)
    assert np.linalg.norm(y.detach().cpu().numpy() - ref) <= 1e-5 * np.linalg.norm(ref)
    np.testing.assert_array_equal(y.detach().cpu().numpy(), l(g, dev(x)).cpu().numpy())      # same as the plain forward
    (y * dev(r)).sum().backward()
    dx, dW, da, db = oracle.grad_gat_conv(s, t, n, x, W, a, b, "relu", r, heads=H, concat=False)
    for got, want in ((xt.grad, dx), (l.dense_x_weight.grad, dW), (l.a.grad, da), (l.bias.grad, db)):
        gn = got.cpu().numpy()
        assert gn.shape == want.shape
        assert np.linalg.norm(gn - want) <= 1e-5 * np.linalg.norm(want)


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din", [(4, 8, 20), (3, 7, 9)])
def test_hip_gatv2_mean_heads_backward_vs_oracle(oracle, H, C, Din):
    import torch
    import gnnmp
    from gnnmp.backward_attn import gatv2_conv_ad
    from gnnmp.layers_attn import GATv2Conv
    from oracle import attn_grads as AG, attn_layers as AL
    gnnmp.load()
    rng = np.random.default_rng(H + C)
    n, E = 1400, 22000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    t[:2500] = 7
    s[3000:5500] = 11
    p = rng.permutation(E)
    s, t = s[p], t[p]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    r = rng.standard_normal((n, C)).astype(np.float32)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    g = gnnmp.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = GATv2Conv((Din, C), "relu", heads=H, concat=False, seed=2)
    l.dense_i_bias = dev((rng.standard_normal(H * C) * 0.1).astype(np.float32))
    l.bias = dev((rng.standard_normal(C) * 0.1).astype(np.float32))
    prm = [l.dense_i_weight, l.dense_i_bias, l.dense_j_weight, l.a, l.bias]
    ref_in = [q.cpu().numpy() for q in prm]
    for q in prm:
        q.requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    y = gatv2_conv_ad(l, g, xt)
    ref = AL.gatv2_conv(s, t, n, x, *ref_in, "relu", heads=H, concat=False)
    assert y.shape == ref.shape == (n, C)
    assert np.linalg.norm(y.detach().cpu().numpy() - ref) <= 1e-5 * np.linalg.norm(ref)
    (y * dev(r)).sum().backward()
    grads = AG.grad_gatv2_conv(s, t, n, x, *ref_in, "relu", r, heads=H, concat=False)
    for got, want in zip([xt.grad] + [q.grad for q in prm], grads):
        gn = got.cpu().numpy()
        assert gn.shape == want.shape
        assert np.linalg.norm(gn - want) <= 1e-5 * np.linalg.norm(want)
