"""Dropout on the attention coefficients — GNNlib/src/layers/conv.jl:139 `α = dropout(α, l.dropout)` — inside the one-pass GAT kernel
(csrc/gat_fused.hip, mode ATTN_GAT_DROP) and its pullback (csrc/gat_backward.hip).  The reference's mask comes from Julia's RNG, so the
draws cannot be matched; what is checked: (1) the mask is the documented function of (seed, edge, head) — known answers computed from
the header's definition with plain Python integers (tests/golden/dropout_keep.json), the oracle's numpy restatement and the HIP code
all agree — and Bernoulli(1 - p); (2) GIVEN that mask, forward and gradients are the reference's formulas (oracle) to 1e-5 (forward) / 1e-5 (gradients: tightened from 3e-5 in round 4)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dropout_keep.json")


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_oracle_mask_matches_the_known_answers(oracle):
    for case in json.load(open(GOLD)):
        if "positions" in case:
            continue
        k = oracle.dropout_keep(case["seed"], case["p"], case["E"], case["H"])
        assert k.tolist() == case["keep"], case


def test_oracle_mask_is_bernoulli(oracle):
    for p in (0.1, 0.5, 0.8):
        k = oracle.dropout_keep(123456789, p, 200000, 4).astype(np.float64)
        n = k.size
        assert abs(k.mean() - (1 - p)) < 5 * np.sqrt(p * (1 - p) / n)
        # heads and neighbouring edges are uncorrelated at the level a sample of this size can see
        assert abs(np.corrcoef(k[:, 0], k[:, 1])[0, 1]) < 0.02 and abs(np.corrcoef(k[:-1, 0], k[1:, 0])[0, 1]) < 0.02
    assert not np.array_equal(oracle.dropout_keep(1, 0.5, 1000, 2), oracle.dropout_keep(2, 0.5, 1000, 2))


def _gat_problem(seed, n=40, E=260, Din=6, H=2, C=4):
    rng = np.random.default_rng(seed)
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    keep = s != t
    s, t = s[keep], t[keep]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    W = (rng.standard_normal((H * C, Din)) / np.sqrt(Din)).astype(np.float32)
    a = (rng.standard_normal((2 * C, H)) * 0.7).astype(np.float32)
    b = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    r = rng.standard_normal((n, H * C)).astype(np.float32)
    return s, t, n, x, W, a, b, r, H


def test_oracle_dropout_forward_is_alpha_times_mask(oracle):
    s, t, n, x, W, a, b, r, H = _gat_problem(3)
    p, seed = 0.4, 987654321
    y0, alpha = oracle.gat_conv(s, t, n, x, W, a, None, None, heads=H, return_alpha=True)
    y1, alpha1 = oracle.gat_conv(s, t, n, x, W, a, None, None, heads=H, return_alpha=True, dropout=p, seed=seed)
    keep = oracle.dropout_keep(seed, p, alpha.shape[0], H)
    assert np.allclose(alpha1, alpha * keep / np.float32(1 - p), rtol=1e-6, atol=0)
    assert (alpha1[keep == 0] == 0).all() and not np.allclose(y0, y1)
    # p = 0 is the undropped layer, bit for bit
    assert np.array_equal(oracle.gat_conv(s, t, n, x, W, a, b, "relu", heads=H, dropout=0.0, seed=5),
                          oracle.gat_conv(s, t, n, x, W, a, b, "relu", heads=H))


@pytest.mark.parametrize("sigma", [None, "relu"])
def test_oracle_dropout_adjoint_matches_finite_differences(oracle, sigma):
    s, t, n, x, W, a, b, r, H = _gat_problem(11)
    p, seed = 0.35, 424242

    def loss(xv, Wv, av, bv):
        y = oracle.gat_conv(s, t, n, xv, Wv, av, bv, sigma, heads=H, dropout=p, seed=seed)
        return float((y.astype(np.float64) * r).sum())

    dx, dW, da, db = oracle.grad_gat_conv(s, t, n, x, W, a, b, sigma, r, heads=H, dropout=p, seed=seed)
    eps = 2e-3          # (dropped coefficients scaled by 1 / (1 - p) put more units near their relu kink than the undropped layer has)
    rng = np.random.default_rng(2)
    args = [x, W, a, b]
    checked = 0
    for which, grad in ((0, dx), (1, dW), (2, da), (3, db)):
        for _ in range(12):
            idx = tuple(int(rng.integers(0, d)) for d in args[which].shape)
            ap = [v.copy() for v in args]
            am = [v.copy() for v in args]
            ap[which][idx] += eps
            am[which][idx] -= eps
            f0, fp, fm = loss(*args), loss(*ap), loss(*am)
            if abs((fp - f0) - (f0 - fm)) > 0.05 * eps * max(1.0, abs(float(grad[idx]))):
                continue                # a relu / leakyrelu kink inside the stencil
            fd = (fp - fm) / (2 * eps)
            assert fd == pytest.approx(float(grad[idx]), rel=3e-2, abs=3e-2)
            checked += 1
    assert checked >= 24


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_kernel_mask_matches_known_answers_and_oracle(gm, oracle):
    from gnnmp import layers
    for case in json.load(open(GOLD)):
        if "positions" in case:
            continue
        k = layers.dropout_keep(case["seed"], case["p"], case["E"], case["H"]).cpu().numpy()
        assert k.tolist() == case["keep"], case
    for seed, p, E, H in ((1, 0.5, 100003, 8), (2**64 - 1, 0.05, 4099, 3), (77, 0.999, 5000, 1), (5, 0.0, 1000, 2)):
        assert np.array_equal(layers.dropout_keep(seed, p, E, H).cpu().numpy(), oracle.dropout_keep(seed, p, E, H)), (seed, p)


def _graph(rng, n, E, hubs=True):
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    if hubs:
        t[: E // 8] = 7                   # hub destination: a split (chunked) row of the forward plan
        s[E // 4: E // 4 + E // 10] = 11   # hub source: a split row of the transposed plan
    p = rng.permutation(E)
    return s[p], t[p]


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din,sigma,concat,loops", [(8, 16, 100, "relu", True, True), (2, 4, 6, None, True, True),
                                                         (1, 64, 32, "relu", True, False), (4, 7, 12, None, True, True),
                                                         (3, 2, 5, "relu", False, True), (1, 1, 3, None, True, True)])
@pytest.mark.parametrize("p", [0.1, 0.6])
def test_forward_vs_oracle_given_the_mask(gm, oracle, H, C, Din, sigma, concat, loops, p):
    import torch
    rng = np.random.default_rng(H * 100 + C)
    n, E = 1500, 24000
    s, t = _graph(rng, n, E)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gm.GATConv((Din, C), sigma, heads=H, concat=concat, add_self_loops=loops, dropout=p, seed=3)
    l.bias = dev((rng.standard_normal(H * C if concat else C) * 0.1).astype(np.float32))
    y = l(g, dev(x))
    seed = l.last_seed
    ref = oracle.gat_conv(s, t, n, x, l.dense_x_weight.cpu().numpy(), l.a.cpu().numpy(), l.bias.cpu().numpy(), sigma, heads=H,
                          concat=concat, add_self_loops_=loops, dropout=p, seed=seed)
    yn = y.cpu().numpy()
    assert np.linalg.norm(yn - ref) <= 1e-5 * np.linalg.norm(ref)
    assert np.abs(yn - ref).max() <= 1e-5 * np.abs(ref).max()
    # a fresh mask per call, as the reference draws one; the same seed reproduces the call bit for bit
    y2 = l(g, dev(x))
    assert l.last_seed != seed and not torch.equal(y2, y)
    from gnnmp.layers import gat_conv
    assert torch.equal(gat_conv(l, g, dev(x), seed=seed), y)
    # and it really drops: against the undropped layer the difference is of the size of the output
    l0 = gm.GATConv((Din, C), sigma, heads=H, concat=concat, add_self_loops=loops, seed=3)
    l0.bias = l.bias
    y0 = l0(g, dev(x))
    assert float((y0 - y).abs().max()) > 1e-3 * float(y0.abs().max())


@pytest.mark.gpu
def test_p_zero_is_the_undropped_kernel(gm):
    import torch
    from gnnmp import _lib as L
    rng = np.random.default_rng(5)
    n, E, H, C = 3000, 40000, 8, 16
    s, t = _graph(rng, n, E)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    plan = g.plan(True)
    Wx = dev(rng.standard_normal((n, H * C)).astype(np.float32))
    a = dev(rng.standard_normal((H, 2 * C)).astype(np.float32))
    o0 = torch.empty((n, H * C), device="cuda"); o1 = torch.empty_like(o0)
    lib = L.load()
    L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, None, 0, L.ptr(o0), H, C, L.stream_ptr()))
    L.check(lib.gnnmp_gat_conv_drop_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, 0.0, 1234, None, 0, L.ptr(o1), None, H, C, L.stream_ptr()))
    assert torch.equal(o0, o1)
    with pytest.raises(L.GnnmpError):
        L.check(lib.gnnmp_gat_conv_drop_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, 1.0, 1, None, 0, L.ptr(o1), None, H, C, L.stream_ptr()))


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din,sigma,concat", [(2, 4, 6, "relu", True), (8, 16, 100, "relu", True), (1, 64, 32, None, True),
                                                  (4, 7, 12, "relu", True), (3, 2, 5, None, False)])
def test_backward_vs_oracle_given_the_mask(gm, oracle, H, C, Din, sigma, concat):
    import torch
    from gnnmp.backward import gat_conv_ad
    rng = np.random.default_rng(H * 1000 + C + 1)
    n, E, p = 1500, 24000, 0.3
    s, t = _graph(rng, n, E)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    nb = H * C if concat else C
    r = rng.standard_normal((n, nb)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gm.GATConv((Din, C), sigma, heads=H, concat=concat, dropout=p, seed=3)
    l.bias = dev((rng.standard_normal(nb) * 0.1).astype(np.float32))
    W0, a0, b0 = l.dense_x_weight.cpu().numpy(), l.a.cpu().numpy(), l.bias.cpu().numpy()
    xt = dev(x).requires_grad_(True)
    for prm in (l.dense_x_weight, l.a, l.bias):
        prm.requires_grad_(True)
    y = gat_conv_ad(l, g, xt)
    seed = l.last_seed
    ref_y = oracle.gat_conv(s, t, n, x, W0, a0, b0, sigma, heads=H, concat=concat, dropout=p, seed=seed)
    assert np.linalg.norm(y.detach().cpu().numpy() - ref_y) <= 1e-5 * np.linalg.norm(ref_y)
    (y * dev(r)).sum().backward()
    dx, dW, da, db = oracle.grad_gat_conv(s, t, n, x, W0, a0, b0, sigma, r, heads=H, concat=concat, dropout=p, seed=seed)
    for name, got, ref in (("dx", xt.grad, dx), ("dW", l.dense_x_weight.grad, dW), ("da", l.a.grad, da), ("db", l.bias.grad, db)):
        gotn = got.cpu().numpy()
        assert gotn.shape == ref.shape, name
        assert np.linalg.norm(gotn - ref) <= 1e-5 * np.linalg.norm(ref), name
    # run-to-run identical with the same seed
    xt2 = dev(x).requires_grad_(True)
    (gat_conv_ad(l, g, xt2, seed=seed) * dev(r)).sum().backward()
    assert bool((xt2.grad == xt.grad).all())


# ---- GATv2Conv (conv.jl:191): the same mask function on the GATv2 logit ----------------------------------------------------------------
def test_oracle_gatv2_dropout_adjoint_matches_finite_differences():
    from oracle import attn_layers as AL, attn_grads as AG
    rng = np.random.default_rng(21)
    n, E, Din, H, C = 40, 260, 6, 2, 4
    s = rng.integers(1, n + 1, E); t = rng.integers(1, n + 1, E)
    keep = s != t
    s, t = s[keep], t[keep]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    Wi = (rng.standard_normal((H * C, Din)) / np.sqrt(Din)).astype(np.float32)
    Wj = (rng.standard_normal((H * C, Din)) / np.sqrt(Din)).astype(np.float32)
    bi = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    a = (rng.standard_normal((C, H)) * 0.7).astype(np.float32)
    b = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    r = rng.standard_normal((n, H * C)).astype(np.float32)
    p, seed = 0.3, 777

    def loss(*v):
        return float((AL.gatv2_conv(s, t, n, v[0], v[1], v[2], v[3], v[4], v[5], None, heads=H, dropout=p, seed=seed).astype(np.float64) * r).sum())

    args = [x, Wi, bi, Wj, a, b]
    grads = AG.grad_gatv2_conv(s, t, n, x, Wi, bi, Wj, a, b, None, r, heads=H, dropout=p, seed=seed)
    eps, checked = 2e-3, 0
    for which, grad in enumerate(grads):
        for _ in range(8):
            idx = tuple(int(rng.integers(0, d)) for d in args[which].shape)
            ap = [v.copy() for v in args]; am = [v.copy() for v in args]
            ap[which][idx] += eps; am[which][idx] -= eps
            f0, fp, fm = loss(*args), loss(*ap), loss(*am)
            if abs((fp - f0) - (f0 - fm)) > 0.05 * eps * max(1.0, abs(float(grad[idx]))):
                continue                # a leakyrelu kink inside the stencil
            assert (fp - fm) / (2 * eps) == pytest.approx(float(grad[idx]), rel=3e-2, abs=3e-2)
            checked += 1
    assert checked >= 36
    # given the mask, the dropped layer is α .* keep ./ (1 - p); p = 0 is the undropped layer bit for bit
    assert np.array_equal(AL.gatv2_conv(s, t, n, x, Wi, bi, Wj, a, b, "relu", heads=H, dropout=0.0, seed=3),
                          AL.gatv2_conv(s, t, n, x, Wi, bi, Wj, a, b, "relu", heads=H))


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din,sigma,concat", [(8, 16, 100, "relu", True), (2, 4, 6, None, True), (1, 64, 32, "relu", True),
                                                  (4, 7, 12, None, True), (3, 2, 5, "relu", False)])
def test_gatv2_dropout_forward_and_backward_vs_oracle(gm, H, C, Din, sigma, concat):
    import torch
    from oracle import attn_layers as AL, attn_grads as AG
    from gnnmp.layers_attn import GATv2Conv, gatv2_conv
    from gnnmp.backward_attn import gatv2_conv_ad
    rng = np.random.default_rng(H * 77 + C)
    n, E, p = 1500, 24000, 0.25
    s, t = _graph(rng, n, E)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    nb = H * C if concat else C
    r = rng.standard_normal((n, nb)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = GATv2Conv((Din, C), sigma, heads=H, concat=concat, dropout=p, seed=4)
    l.dense_i_bias = dev((rng.standard_normal(H * C) * 0.1).astype(np.float32))
    l.bias = dev((rng.standard_normal(nb) * 0.1).astype(np.float32))
    prm = [l.dense_i_weight, l.dense_i_bias, l.dense_j_weight, l.a, l.bias]
    ref_in = [q.cpu().numpy() for q in prm]
    # forward (inference path)
    y = l(g, dev(x))
    seed = l.last_seed
    ref = AL.gatv2_conv(s, t, n, x, *ref_in, sigma, heads=H, concat=concat, dropout=p, seed=seed)
    assert np.linalg.norm(y.cpu().numpy() - ref) <= 1e-5 * np.linalg.norm(ref)
    assert torch.equal(gatv2_conv(l, g, dev(x), seed=seed), y) and not torch.equal(l(g, dev(x)), y)
    # forward + pullback through autograd with the same seed
    for q in prm:
        q.requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    ya = gatv2_conv_ad(l, g, xt, seed=seed)
    assert np.linalg.norm(ya.detach().cpu().numpy() - ref) <= 1e-5 * np.linalg.norm(ref)
    (ya * dev(r)).sum().backward()
    grads = AG.grad_gatv2_conv(s, t, n, x, *ref_in, sigma, r, heads=H, concat=concat, dropout=p, seed=seed)
    for name, got, gref in zip(("dx", "dWi", "dbi", "dWj", "da", "db"), (xt.grad, *(q.grad for q in prm)), grads):
        gotn = got.cpu().numpy()
        assert gotn.shape == gref.shape, name
        if np.linalg.norm(gref) < 1e-6:
            assert np.linalg.norm(gotn) <= 1e-3, name
        else:
            assert np.linalg.norm(gotn - gref) <= 1e-5 * np.linalg.norm(gref), name
