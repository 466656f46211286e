"""Adjoints.  CPU: the oracle's restated NNlib rrules against central finite differences in
float64.  GPU: the HIP adjoints (transposed-plan propagate, edge-dot, max/min gradient) against the oracle, and through
torch.autograd."""
import numpy as np
import pytest


def _problem(seed=0, n=60, E=500, D=5):
    rng = np.random.default_rng(seed)
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 5, E)
    x = rng.standard_normal((n, D)).astype(np.float32)
    w = (rng.random(E) + 0.5).astype(np.float32)
    r = rng.standard_normal((n, D)).astype(np.float32)
    return s, t, n, x, w, r


@pytest.mark.parametrize("aggr", ["+", "mean", "max", "min"])
@pytest.mark.parametrize("weighted", [False, True])
def test_oracle_adjoint_matches_finite_differences(oracle, aggr, weighted):
    if weighted and aggr in ("max", "min"):
        pytest.skip("weighted max/min is not on the path")
    s, t, n, x, w, r = _problem()
    if aggr in ("max", "min"):
        # NNlib's rule hands Δ to EVERY tied maximiser, so a multi-edge counts twice — a subgradient convention that a
        # finite difference cannot see: compare on a simple graph
        _, keep = np.unique(s * 1000 + t, return_index=True)
        keep = np.sort(keep)
        s, t, w = s[keep], t[keep], w[keep]
    ww = w if weighted else None
    has = np.bincount(t - 1, minlength=n) > 0          # empty destinations hold -Inf / +Inf under max / min
    r = r * has[:, None]

    def loss(xv, wv):
        y = oracle.propagate(aggr, s, t, n, xv.astype(np.float32), None if wv is None else wv.astype(np.float32))
        return float((y[has].astype(np.float64) * r[has]).sum())

    dx, dw = oracle.grad_propagate(aggr, s, t, n, r, x, ww)
    eps = 1e-2
    rng = np.random.default_rng(1)
    for _ in range(25):
        i, d = int(rng.integers(0, n)), int(rng.integers(0, x.shape[1]))
        xp, xm = x.copy(), x.copy()
        xp[i, d] += eps
        xm[i, d] -= eps
        f0, fp, fm = loss(x, ww), loss(xp, ww), loss(xm, ww)
        if abs((fp - f0) - (f0 - fm)) > 1e-3 * eps * max(1.0, abs(float(dx[i, d]))) * 50:
            continue                                   # a kink of max/min inside [x - eps, x + eps]: not differentiable here
        fd = (fp - fm) / (2 * eps)
        assert fd == pytest.approx(float(dx[i, d]), rel=2e-2, abs=2e-2)
    if weighted:
        for _ in range(25):
            k = int(rng.integers(0, len(s)))
            wp, wm = w.copy(), w.copy()
            wp[k] += eps
            wm[k] -= eps
            fd = (loss(x, wp) - loss(x, wm)) / (2 * eps)
            assert fd == pytest.approx(float(dw[k]), rel=2e-2, abs=2e-2)


# ---------------------------------------------------------------------------------------------------------------
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
@pytest.mark.parametrize("D", [1, 5, 8, 16, 36, 100, 128, 200, 250])      # edge_dot lane groups of 1, 8, 2, 4, 16, 32, 32, 64; COO fallback
def test_hip_adjoints_vs_oracle(gm, oracle, D):
    from gnnmp import backward as bw
    rng = np.random.default_rng(D)
    n, E = 900, 20000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 10, E)
    s[:1500] = 4                     # a source with 1500 outgoing edges: split row of the TRANSPOSED plan
    t[2000:3200] = 9                 # and a hub destination
    p = rng.permutation(E)
    s, t = s[p], t[p]
    x = rng.standard_normal((n, D)).astype(np.float32)
    x[rng.integers(0, n, 50)] = x[0]                       # exact ties for max/min
    w = (rng.random(E) + 0.5).astype(np.float32)
    dy = rng.standard_normal((n, D)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    outdeg = np.bincount(s - 1, minlength=n)
    short = outdeg <= 64

    def check(got, ref, exact_rows=short):
        np.testing.assert_array_equal(got[exact_rows], ref[exact_rows])
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= 4e-5 * scale

    # Δxj, aggr = + (bit-exact where the source's out-degree is not split), with and without weights
    dx, _ = oracle.grad_propagate("+", s, t, n, dy, x)
    check(bw.propagate_grad_xj(g, "+", dev(dy)).cpu().numpy(), dx)
    dxw, dw = oracle.grad_propagate("+", s, t, n, dy, x, w)
    check(bw.propagate_grad_xj(g, "+", dev(dy), w=dev(w)).cpu().numpy(), dxw)
    for coo in (False, True):            # destination-sorted walk (plan) and plain COO-order kernel
        got_dw = bw.propagate_grad_w(g, dev(dy), dev(x), coo_order=coo).cpu().numpy()
        assert np.abs(got_dw - dw).max() <= 1e-5 * np.abs(dw).max() * 4
    # mean: multiply-by-reciprocal vs NNlib's division: tolerance
    dxm, _ = oracle.grad_propagate("mean", s, t, n, dy, x)
    got = bw.propagate_grad_xj(g, "mean", dev(dy)).cpu().numpy()
    assert np.linalg.norm(got - dxm) <= 1e-5 * np.linalg.norm(dxm)
    # max / min (ties all receive the gradient)
    for aggr in ("max", "min"):
        y = gm.propagate(gm.copy_xj, g, aggr, xj=dev(x))
        ref, _ = oracle.grad_propagate(aggr, s, t, n, dy, x)
        check(bw.propagate_grad_xj(g, aggr, dev(dy), xj=dev(x), y=y).cpu().numpy(), ref)


@pytest.mark.gpu
def test_autograd_through_hip_propagate(gm, oracle):
    import torch
    from gnnmp.backward import propagate_ad
    rng = np.random.default_rng(3)
    n, E, D = 400, 6000, 12
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    x = rng.standard_normal((n, D)).astype(np.float32)
    w = (rng.random(E) + 0.5).astype(np.float32)
    r = rng.standard_normal((n, D)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    for aggr, weighted in (("+", True), ("mean", True), ("+", False), ("max", False), ("min", False)):
        xt = dev(x).requires_grad_(True)
        wt = dev(w).requires_grad_(True) if weighted else None
        y = propagate_ad(g, aggr, xt, wt)
        (y * dev(r)).sum().backward()
        dx, dw = oracle.grad_propagate(aggr, s, t, n, r, x, w if weighted else None)
        assert np.linalg.norm(xt.grad.cpu().numpy() - dx) <= 1e-5 * np.linalg.norm(dx)
        if weighted:
            assert np.linalg.norm(wt.grad.cpu().numpy() - dw) <= 1e-5 * np.linalg.norm(dw)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,Dout", [(1000, 100, 100), (4097, 37, 130), (300, 128, 16), (5, 3, 2), (70000, 16, 128),
                                      (2050, 200, 256), (1027, 300, 40), (999, 64, 129), (3, 100, 100), (40001, 100, 128)])
def test_dense_adjoints_vs_float64(gm, N, K, Dout):
    from gnnmp import backward as bw
    rng = np.random.default_rng(N + K)
    x = rng.standard_normal((N, K)).astype(np.float32)
    W = (rng.standard_normal((Dout, K)) / np.sqrt(K)).astype(np.float32)
    y = rng.standard_normal((N, Dout)).astype(np.float32)
    dy = rng.standard_normal((N, Dout)).astype(np.float32)
    dz = bw.act_grad(dev(dy), dev(y), "relu")
    ref_dz = np.where(y > 0, dy, 0).astype(np.float32)
    np.testing.assert_array_equal(dz.cpu().numpy(), ref_dz)
    dW, db = bw.dense_grad_w(dz, dev(x))
    ref_dW = ref_dz.astype(np.float64).T @ x.astype(np.float64)
    ref_db = ref_dz.astype(np.float64).sum(0)
    assert np.linalg.norm(dW.cpu().numpy() - ref_dW) <= 1e-5 * np.linalg.norm(ref_dW)
    assert np.linalg.norm(db.cpu().numpy() - ref_db) <= 1e-5 * max(np.linalg.norm(ref_db), 1e-3 * np.sqrt(N))
    dx = bw.dense_grad_x(dz, dev(W))
    ref_dx = ref_dz.astype(np.float64) @ W.astype(np.float64)
    assert np.linalg.norm(dx.cpu().numpy() - ref_dx) <= 1e-5 * np.linalg.norm(ref_dx)
    # deterministic (no atomics)
    dW2, db2 = bw.dense_grad_w(dz, dev(x))
    assert bool((dW2 == dW).all()) and bool((db2 == db).all())
    # the round-1 32x32x2 kernel stays behind knob 10 < 0 (A/B runs): same product
    gm.tune(10, -1)
    try:
        dW3, _ = bw.dense_grad_w(dz, dev(x))
    finally:
        gm.tune(10, 0)
    assert np.linalg.norm(dW3.cpu().numpy() - ref_dW) <= 1e-5 * np.linalg.norm(ref_dW)


@pytest.mark.gpu
@pytest.mark.parametrize("Din,Dout", [(24, 24), (16, 40), (40, 8)])
def test_gcn_layer_backward_vs_oracle(gm, oracle, Din, Dout):
    import torch
    from gnnmp.backward import gcn_conv_ad
    rng = np.random.default_rng(Din * 100 + Dout)
    n, E = 700, 9000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    x = rng.standard_normal((n, Din)).astype(np.float32)
    r = rng.standard_normal((n, Dout)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gm.GCNConv((Din, Dout), "relu", seed=5)
    l.bias = dev(rng.standard_normal(Dout).astype(np.float32) * 0.1)
    W0, b0 = l.weight.cpu().numpy(), l.bias.cpu().numpy()
    xt = dev(x).requires_grad_(True)
    l.weight.requires_grad_(True)
    l.bias.requires_grad_(True)
    y = gcn_conv_ad(l, g, xt)
    ref_y = oracle.gcn_conv(s, t, n, x, W0, b0, "relu")
    assert np.linalg.norm(y.detach().cpu().numpy() - ref_y) <= 1e-5 * np.linalg.norm(ref_y)
    (y * dev(r)).sum().backward()
    dx, dW, db = oracle.grad_gcn_conv(s, t, n, x, W0, b0, "relu", r)
    for got, ref in ((xt.grad, dx), (l.weight.grad, dW), (l.bias.grad, db)):
        gotn = got.cpu().numpy()
        assert np.linalg.norm(gotn - ref) <= 1e-5 * np.linalg.norm(ref)


# ---------------------------------------------------------------------------------------------------------------
# GATConv
def _gat_problem(seed, n=40, E=260, Din=6, H=2, C=4):
    rng = np.random.default_rng(seed)
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    keep = s != t                      # GATConv removes nothing itself; keep the self-looped graph simple
    s, t = s[keep], t[keep]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    W = (rng.standard_normal((H * C, Din)) / np.sqrt(Din)).astype(np.float32)
    a = (rng.standard_normal((2 * C, H)) * 0.7).astype(np.float32)
    b = (rng.standard_normal(H * C) * 0.1).astype(np.float32)
    r = rng.standard_normal((n, H * C)).astype(np.float32)
    return s, t, n, x, W, a, b, r, H


@pytest.mark.parametrize("sigma", [None, "relu"])
def test_oracle_gat_adjoint_matches_finite_differences(oracle, sigma):
    s, t, n, x, W, a, b, r, H = _gat_problem(11)

    def loss(xv, Wv, av, bv):
        y = oracle.gat_conv(s, t, n, xv, Wv, av, bv, sigma, heads=H)
        return float((y.astype(np.float64) * r).sum())

    dx, dW, da, db = oracle.grad_gat_conv(s, t, n, x, W, a, b, sigma, r, heads=H)
    eps = 1e-2
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
    assert checked >= 30


@pytest.mark.gpu
@pytest.mark.parametrize("H,C,Din,sigma", [(2, 4, 6, "relu"), (8, 16, 100, "relu"), (1, 64, 32, None), (4, 8, 20, None),
                                          (3, 2, 5, "relu"), (1, 1, 3, None), (4, 7, 12, "relu"), (2, 12, 9, None)])
def test_gat_layer_backward_vs_oracle(gm, oracle, H, C, Din, sigma):
    import torch
    from gnnmp.backward import gat_conv_ad
    rng = np.random.default_rng(H * 1000 + C)
    n, E = 1500, 24000
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n + 1, E)
    t[:3000] = 7                      # hub destination: split row of the forward plan
    s[5000:7500] = 11                 # hub source: split row of the transposed plan
    p = rng.permutation(E)
    s, t = s[p], t[p]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    r = rng.standard_normal((n, H * C)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gm.GATConv((Din, C), sigma, heads=H, seed=3)
    l.bias = dev((rng.standard_normal(H * C) * 0.1).astype(np.float32))
    W0, a0, b0 = l.dense_x_weight.cpu().numpy(), l.a.cpu().numpy(), l.bias.cpu().numpy()
    xt = dev(x).requires_grad_(True)
    for prm in (l.dense_x_weight, l.a, l.bias):
        prm.requires_grad_(True)
    y = gat_conv_ad(l, g, xt)
    ref_y = oracle.gat_conv(s, t, n, x, W0, a0, b0, sigma, heads=H)
    assert np.linalg.norm(y.detach().cpu().numpy() - ref_y) <= 1e-5 * np.linalg.norm(ref_y)
    (y * dev(r)).sum().backward()
    dx, dW, da, db = oracle.grad_gat_conv(s, t, n, x, W0, a0, b0, sigma, r, heads=H)
    for name, got, ref in (("dx", xt.grad, dx), ("dW", l.dense_x_weight.grad, dW), ("da", l.a.grad, da),
                           ("db", l.bias.grad, db)):
        gotn = got.cpu().numpy()
        assert gotn.shape == ref.shape, name
        assert np.linalg.norm(gotn - ref) <= 1e-5 * np.linalg.norm(ref), name
    # run-to-run identical (no atomics anywhere on the path)
    xt2 = dev(x).requires_grad_(True)
    l.dense_x_weight.grad = None
    (gat_conv_ad(l, g, xt2) * dev(r)).sum().backward()
    assert bool((xt2.grad == xt.grad).all())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,aggr", [("graph", "+"), ("graph", "max"), ("sage", "mean"), ("sage", "+")])
def test_graph_and_sage_layer_backward_vs_oracle(gm, oracle, kind, aggr):
    import torch
    from gnnmp.backward import graph_conv_ad, sage_conv_ad
    rng = np.random.default_rng(len(kind) * 7 + len(aggr))
    n, E, Din, Dout = 800, 9000, 20, 36
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 3, E)              # the last nodes receive nothing (empty rows: 0 under + / mean)
    if aggr == "max":
        # every node gets an in-edge (an empty row holds -Inf and 0 * -Inf = NaN in ΔW2, in the reference too), and the
        # graph is simple (NNlib's tie rule hands Δ to both copies of a multi-edge)
        s = np.concatenate([s, np.arange(1, n + 1)])
        t = np.concatenate([t, np.roll(np.arange(1, n + 1), 1)])
        _, keep = np.unique(s * 10000 + t, return_index=True)
        s, t = s[np.sort(keep)], t[np.sort(keep)]
    x = rng.standard_normal((n, Din)).astype(np.float32)
    r = rng.standard_normal((n, Dout)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    if kind == "graph":
        l = gm.GraphConv((Din, Dout), "relu", aggr=aggr, seed=4)
        params = [l.weight1, l.weight2, l.bias]
        W1, W2 = l.weight1.cpu().numpy(), l.weight2.cpu().numpy()
        fn = graph_conv_ad
    else:
        l = gm.SAGEConv((Din, Dout), "relu", aggr=aggr, seed=4)
        params = [l.weight, l.bias]
        Wn = l.weight.cpu().numpy()
        W1, W2 = Wn[:, :Din], Wn[:, Din:]
        fn = sage_conv_ad
    l.bias = dev((rng.standard_normal(Dout) * 0.1).astype(np.float32))
    params[-1] = l.bias
    b0 = l.bias.cpu().numpy()
    for prm in params:
        prm.requires_grad_(True)
    xt = dev(x).requires_grad_(True)
    y = fn(l, g, xt)
    (y * dev(r)).sum().backward()
    dx, dW1, dW2, db = oracle.grad_graph_conv(s, t, n, x, W1, W2, b0, "relu", r, aggr=aggr)
    got_w = [params[0].grad.cpu().numpy(), params[1].grad.cpu().numpy()] if kind == "graph" else \
        [params[0].grad.cpu().numpy()[:, :Din], params[0].grad.cpu().numpy()[:, Din:]]
    for name, got, ref in (("dx", xt.grad.cpu().numpy(), dx), ("dW1", got_w[0], dW1), ("dW2", got_w[1], dW2),
                           ("db", l.bias.grad.cpu().numpy(), db)):
        assert np.isfinite(got).all(), name
        assert np.linalg.norm(got - ref) <= 1e-5 * np.linalg.norm(ref), name


@pytest.mark.gpu
@pytest.mark.parametrize("aggr", ["+", "mean"])
def test_global_pool_pullback(gm, aggr):
    """∇reduce_nodes: NNlib's rule for scatter(+ | mean, x, graph_indicator) — Δx = gather(Δ, idx) (./ count[idx]) — through the
    autograd wrapper, against the same two lines in numpy (float64); graphs of 1..40 nodes, D not a multiple of 4"""
    import torch
    from gnnmp import synth
    from gnnmp.backward import global_pool_ad
    rng = np.random.default_rng(5)
    members = []
    for _ in range(300):
        n = int(rng.integers(1, 41))
        members.append((np.zeros(0, np.int64), np.zeros(0, np.int64), n))
    D = 7
    xs = [rng.standard_normal((n, D)).astype(np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    x = g.x.clone().requires_grad_(True)
    y = global_pool_ad(gm.GlobalPool(aggr), g, x)
    r = rng.standard_normal((len(members), D)).astype(np.float32)
    (y * dev(r)).sum().backward()
    sizes = np.array([n for _, _, n in members])
    gi = np.repeat(np.arange(len(members)), sizes)
    ref_y = np.zeros((len(members), D))
    np.add.at(ref_y, gi, np.concatenate(xs).astype(np.float64))
    if aggr == "mean":
        ref_y /= sizes[:, None]
    ref_dx = r.astype(np.float64)[gi] / (sizes[gi][:, None] if aggr == "mean" else 1.0)
    assert np.abs(y.detach().cpu().numpy() - ref_y).max() <= 1e-5 * np.abs(ref_y).max()
    assert np.abs(x.grad.cpu().numpy() - ref_dx).max() <= 1e-6 * np.abs(ref_dx).max()
