"""Pullbacks of GATv2Conv and TransformerConv.  CPU: the oracle's rule-by-rule adjoints (oracle/attn_grads.py) against
central finite differences of the oracle forward.  GPU: the two-pass HIP pullback (gnnmp_attn_conv_grad_f32) through
torch.autograd against the oracle, on graphs whose hubs split rows on both plans."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def AL(oracle):
    from oracle import attn_layers
    return attn_layers


@pytest.fixture(scope="module")
def AG(oracle):
    from oracle import attn_grads
    return attn_grads


def _graph(rng, n, E):
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    k = s != t
    return s[k], t[k]


def _fd_check(loss, args, grads, rng, per=10, eps=2e-3, need=25):
    checked = 0
    for which, grad in grads:
        for _ in range(per):
            idx = tuple(int(rng.integers(0, d)) for d in args[which].shape)
            ap = [None if v is None else v.copy() for v in args]
            am = [None if v is None else v.copy() for v in args]
            ap[which][idx] += eps
            am[which][idx] -= eps
            f0, fp, fm = loss(*args), loss(*ap), loss(*am)
            if abs((fp - f0) - (f0 - fm)) > 0.05 * eps * max(1.0, abs(float(grad[idx]))):
                continue                                   # a relu / leakyrelu kink inside the stencil
            assert (fp - fm) / (2 * eps) == pytest.approx(float(grad[idx]), rel=3e-2, abs=3e-2)
            checked += 1
    assert checked >= need


@pytest.mark.parametrize("sigma", [None, "relu"])
def test_oracle_gatv2_adjoint_vs_finite_differences(oracle, AL, AG, sigma):
    rng = np.random.default_rng(31)
    n, Din, H, C = 40, 5, 2, 4
    s, t = _graph(rng, n, 260)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    Wi = (rng.standard_normal((H * C, Din)) * 0.5).astype(np.float32)
    bi = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    Wj = (rng.standard_normal((H * C, Din)) * 0.5).astype(np.float32)
    a = (rng.standard_normal((C, H)) * 0.7).astype(np.float32)
    b = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    r = rng.standard_normal((n, H * C)).astype(np.float32)

    def loss(xv, Wiv, biv, Wjv, av, bv):
        return float((AL.gatv2_conv(s, t, n, xv, Wiv, biv, Wjv, av, bv, sigma, heads=H).astype(np.float64) * r).sum())

    g = AG.grad_gatv2_conv(s, t, n, x, Wi, bi, Wj, a, b, sigma, r, heads=H)
    _fd_check(loss, [x, Wi, bi, Wj, a, b], list(enumerate(g)), np.random.default_rng(1), per=8, need=35)


@pytest.mark.parametrize("concat", [True, False])
def test_oracle_transformer_adjoint_vs_finite_differences(oracle, AL, AG, concat):
    rng = np.random.default_rng(32)
    n, H, C = 36, 2, 3
    Din = Dout = H * C if concat else C
    s, t = _graph(rng, n, 220)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    mk = lambda rr: ((rng.standard_normal((rr, Din)) * 0.5).astype(np.float32), (rng.standard_normal(rr) * 0.1).astype(np.float32))
    (W1, b1), (W2, b2), (W3, b3), (W4, b4) = mk(Dout), mk(H * C), mk(H * C), mk(H * C)
    r = rng.standard_normal((n, Dout)).astype(np.float32)

    def loss(xv, W1v, b1v, W2v, b2v, W3v, b3v, W4v, b4v):
        y = AL.transformer_conv(s, t, n, xv, W1v, b1v, W2v, b2v, W3v, b3v, W4v, b4v, heads=H, concat=concat,
                                add_self_loops_=True, skip_connection=True)
        return float((y.astype(np.float64) * r).sum())

    dx, gw = AG.grad_transformer_conv(s, t, n, x, W1, b1, W2, b2, W3, b3, W4, b4, r, heads=H, add_self_loops_=True,
                                      skip_connection=True, concat=concat)
    grads = [(0, dx)] + [(1 + 2 * k, gw[f"W{k + 1}"]) for k in range(4)] + [(2 + 2 * k, gw[f"b{k + 1}"]) for k in range(4)]
    _fd_check(loss, [x, W1, b1, W2, b2, W3, b3, W4, b4], grads, np.random.default_rng(2), per=6, need=40)


def test_oracle_agnn_adjoint_vs_finite_differences(oracle, AL, AG):
    rng = np.random.default_rng(41)
    n, D = 30, 5
    s, t = _graph(rng, n, 180)
    x = rng.standard_normal((n, D)).astype(np.float32)
    r = rng.standard_normal((n, D)).astype(np.float32)
    beta = np.array([1.7], np.float32)

    def loss(xv, bv):
        return float((AL.agnn_conv(s, t, n, xv, float(bv[0])).astype(np.float64) * r).sum())

    dx, db = AG.grad_agnn_conv(s, t, n, x, float(beta[0]), r)
    _fd_check(loss, [x, beta], [(0, dx), (1, np.array([db], np.float32))], np.random.default_rng(4), per=20, need=15)


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
    t[: E // 8] = 7                       # hub destination: split row of the forward plan
    s[E // 4: E // 4 + E // 10] = 11      # hub source: split row of the transposed plan
    p = rng.permutation(E)
    return s[p], t[p]


def close(got, ref, tol=1e-5):
    got = got.detach().cpu().numpy() if hasattr(got, "detach") else got
    assert got.shape == ref.shape
    if np.linalg.norm(ref) < 1e-6:
        # a gradient that vanishes identically (the bias of the keys shifts every logit of a neighbourhood by the same
        # Q_i . b: softmax does not see it): the device sums N cancelling fp32 terms of size O(1)
        assert np.linalg.norm(got) <= 1e-3
        return
    assert np.linalg.norm(got - ref) <= tol * np.linalg.norm(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din,sigma", [(2, 4, 6, "relu"), (8, 16, 100, "relu"), (1, 64, 32, None), (4, 8, 20, None), (1, 3, 5, None),
                                          (4, 7, 12, "relu"), (3, 5, 6, None)])
def test_hip_gatv2_backward_vs_oracle(gm, AL, AG, H, C, Din, sigma):
    from gnnmp.backward_attn import gatv2_conv_ad
    from gnnmp.layers_attn import GATv2Conv
    rng = np.random.default_rng(H * 50 + C)
    n, E = 1500, 24000
    s, t = hub_graph(rng, n, E)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    r = rng.standard_normal((n, H * C)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = GATv2Conv((Din, C), sigma, heads=H, seed=4)
    l.dense_i_bias = dev((rng.standard_normal(H * C) * 0.1).astype(np.float32))
    l.bias = dev((rng.standard_normal(H * C) * 0.1).astype(np.float32))
    prm = [l.dense_i_weight, l.dense_i_bias, l.dense_j_weight, l.a, l.bias]
    ref_in = [p.cpu().numpy() for p in prm]
    for p in prm:
        p.requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    y = gatv2_conv_ad(l, g, xt)
    ref_y = AL.gatv2_conv(s, t, n, x, ref_in[0], ref_in[1], ref_in[2], ref_in[3], ref_in[4], sigma, heads=H)
    close(y, ref_y, 1e-5)
    (y * dev(r)).sum().backward()
    dx, dWi, dbi, dWj, da, db = AG.grad_gatv2_conv(s, t, n, x, ref_in[0], ref_in[1], ref_in[2], ref_in[3], ref_in[4], sigma, r,
                                                   heads=H)
    for name, got, ref in (("dx", xt.grad, dx), ("dWi", prm[0].grad, dWi), ("dbi", prm[1].grad, dbi), ("dWj", prm[2].grad, dWj),
                           ("da", prm[3].grad, da), ("db", prm[4].grad, db)):
        close(got, ref)
    xt2 = dev(x).requires_grad_(True)
    (gatv2_conv_ad(l, g, xt2) * dev(r)).sum().backward()
    assert bool((xt2.grad == xt.grad).all())              # no atomics: run-to-run identical


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,root,skip,loops,concat", [(2, 4, True, True, True, True), (8, 16, True, False, False, True),
                                                       (1, 32, False, True, True, True), (4, 8, True, False, True, True),
                                                       (4, 8, True, True, True, False), (3, 5, False, False, False, False)])
def test_hip_transformer_backward_vs_oracle(gm, AL, AG, H, C, root, skip, loops, concat):
    from gnnmp.backward_attn import transformer_conv_ad
    from gnnmp.layers_attn import TransformerConv
    rng = np.random.default_rng(H * 9 + C)
    n, E = 1400, 22000
    s, t = hub_graph(rng, n, E)
    Dout = H * C if concat else C
    Din = Dout if skip else 20
    x = rng.standard_normal((n, Din)).astype(np.float32)
    r = rng.standard_normal((n, Dout)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = TransformerConv((Din, C), heads=H, concat=concat, add_self_loops=loops, root_weight=root, skip_connection=skip, seed=6)
    names = ["W1", "W2", "W3", "W4"] if root else ["W2", "W3", "W4"]
    for nm in names:
        setattr(l, f"{nm}_bias", dev((rng.standard_normal(getattr(l, f"{nm}_bias").numel()) * 0.1).astype(np.float32)))
        getattr(l, f"{nm}_weight").requires_grad_(True)
        getattr(l, f"{nm}_bias").requires_grad_(True)
    c = lambda v: None if v is None else v.detach().cpu().numpy()
    w = {nm: (c(getattr(l, f"{nm}_weight")), c(getattr(l, f"{nm}_bias"))) for nm in ("W1", "W2", "W3", "W4")}
    xt = dev(x).requires_grad_(True)
    y = transformer_conv_ad(l, g, xt)
    args = (w["W1"][0], w["W1"][1], w["W2"][0], w["W2"][1], w["W3"][0], w["W3"][1], w["W4"][0], w["W4"][1])
    close(y, AL.transformer_conv(s, t, n, x, *args, heads=H, concat=concat, add_self_loops_=loops, skip_connection=skip), 1e-5)
    (y * dev(r)).sum().backward()
    dx, gw = AG.grad_transformer_conv(s, t, n, x, *args, r, heads=H, add_self_loops_=loops, skip_connection=skip, concat=concat)
    close(xt.grad, dx)
    for nm in names:
        k = nm[1]
        close(getattr(l, f"{nm}_weight").grad, gw[f"W{k}"])
        close(getattr(l, f"{nm}_bias").grad, gw[f"b{k}"])


@pytest.mark.gpu
@pytest.mark.parametrize("D,beta,loops", [(16, 1.0, True), (100, 2.5, True), (7, 0.6, False), (64, -1.3, True)])
def test_hip_agnn_backward_vs_oracle(gm, AL, AG, D, beta, loops):
    import torch
    from gnnmp.backward_attn import agnn_conv_ad
    from gnnmp.layers_attn import AGNNConv
    rng = np.random.default_rng(D)
    n, E = 1500, 24000
    s, t = hub_graph(rng, n, E)
    if not loops:                                          # every node needs an in-edge (softmax over an empty set otherwise)
        s = np.concatenate([s, np.roll(np.arange(1, n + 1), 1)])
        t = np.concatenate([t, np.arange(1, n + 1)])
    x = rng.standard_normal((n, D)).astype(np.float32)
    r = rng.standard_normal((n, D)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = AGNNConv(init_beta=beta, add_self_loops=loops)
    l.beta = torch.tensor([beta], dtype=torch.float32, device="cuda", requires_grad=True)
    xt = dev(x).requires_grad_(True)
    y = agnn_conv_ad(l, g, xt)
    ref = AL.agnn_conv(s, t, n, x, beta, add_self_loops_=loops)
    close(y, ref, 1e-5)
    plain = AGNNConv(init_beta=beta, add_self_loops=loops)(g, dev(x))       # the one-pass cosine kernel: same layer
    close(plain, ref, 1e-5)
    (y * dev(r)).sum().backward()
    dx, db = AG.grad_agnn_conv(s, t, n, x, beta, r, add_self_loops_=loops)
    close(xt.grad, dx)
    assert abs(float(l.beta.grad[0]) - float(db)) <= 1e-4 * max(1.0, abs(float(db)))
    xt2 = dev(x).requires_grad_(True)
    (agnn_conv_ad(l, g, xt2) * dev(r)).sum().backward()
    assert bool((xt2.grad == xt.grad).all())


@pytest.mark.gpu
def test_attn_grad_rejects_the_cosine_logit(gm):
    import torch
    from gnnmp import _lib as L
    g = gm.GNNGraph(dev(np.array([1, 2])), dev(np.array([2, 1])), num_nodes=2)
    from gnnmp.backward import plan_transposed
    z = torch.zeros((2, 4), device="cuda")
    st = torch.zeros((2, 1, 2), device="cuda")
    ln = torch.zeros((2, 1, 4), device="cuda")
    rc = L.load().gnnmp_attn_conv_grad_f32(g.plan(True).handle, plan_transposed(g, True).handle, 3, L.ptr(z), L.ptr(z), None, None,
                                           0.2, 1.0, L.ptr(st), L.ptr(z), L.ptr(ln), L.ptr(z), L.ptr(z), None, None, None, 1, 4,
                                           L.stream_ptr())
    assert rc == L.EUNSUPPORTED
