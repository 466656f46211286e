"""gnnmp_gat_conv_train_f32 / gnnmp_gat_conv_grad2_f32 (round 4): the GAT pullback with ONE edge pass.  The training forward's extra
outputs are what their definitions say — o+_i = Σ_{j: z_ij > 0} α_ij Wx_j, P_i = Σ_{j: z_ij > 0} α_ij (float64 edge loops) — its
out / stats are bit-identical to gnnmp_gat_conv_stats_f32's, and the new pullback agrees with the two-pass one
(gnnmp_gat_conv_grad_f32) on graphs with split rows on both plans, isolated nodes, relu and identity tails, several head shapes."""
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


def graph(gm, N, E, seed, hubs=True):
    import torch
    rng = np.random.default_rng(seed)
    s = rng.integers(0, N, E)
    t = rng.integers(0, N - 5, E)                     # the last 5 nodes have no in-edge
    if hubs:                                          # rows longer than the long-row threshold on both plans
        s = np.concatenate([s, rng.integers(0, N, 700), np.full(900, 3)])
        t = np.concatenate([t, np.full(700, 1), rng.integers(0, N - 5, 900)])
    return s.astype(np.int64) + 1, t.astype(np.int64) + 1


@pytest.mark.parametrize("H,C,act,bias", [(8, 16, "relu", True), (4, 8, None, False), (1, 32, "relu", True), (2, 4, None, True)])
def test_train_forward_and_one_pass_pullback(gm, H, C, act, bias):
    import torch
    from gnnmp import _lib as L
    from gnnmp.backward import plan_transposed
    lib = L.load()
    N, E = 3000, 40000
    s, t = graph(gm, N, E, seed=H * 10 + C)
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N)
    plan, plan_t = g.plan(False), plan_transposed(g, False)
    assert plan.n_long > 0 and plan_t.n_long > 0
    gen = torch.Generator(device="cpu"); gen.manual_seed(5)
    D = H * C
    Wx = (torch.randn((N, D), generator=gen) * 0.7).cuda()
    a = (torch.randn((H, 2 * C), generator=gen) * 0.5).cuda()
    b = (torch.randn(D, generator=gen) * 0.3).cuda() if bias else None
    code = L.ACT_RELU if act == "relu" else L.ACT_IDENTITY
    f32 = dict(dtype=torch.float32, device="cuda")
    out0, st0 = torch.empty((N, D), **f32), torch.empty((N, H, 2), **f32)
    L.check(lib.gnnmp_gat_conv_stats_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, L.ptr(b), code, L.ptr(out0), L.ptr(st0), H, C, L.stream_ptr()))
    out1, st1 = torch.empty((N, D), **f32), torch.empty((N, H, 2), **f32)
    op, pp = torch.empty((N, D), **f32), torch.empty((N, H), **f32)
    L.check(lib.gnnmp_gat_conv_train_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, L.ptr(b), code, L.ptr(out1), L.ptr(st1), L.ptr(op),
                                         L.ptr(pp), H, C, L.stream_ptr()))
    assert torch.equal(out0, out1) and torch.equal(st0, st1)
    # o+ and P against float64 edge loops
    w = Wx.double().cpu().numpy().reshape(N, H, C)
    an = a.double().cpu().numpy()
    sd = np.einsum("nhc,hc->nh", w, an[:, :C]); ss = np.einsum("nhc,hc->nh", w, an[:, C:])
    z = sd[t - 1] + ss[s - 1]
    l = np.where(z > 0, z, 0.2 * z)
    mx = np.full((N, H), -np.inf); np.maximum.at(mx, t - 1, l)
    ex = np.exp(l - mx[t - 1])
    den = np.zeros((N, H)); np.add.at(den, t - 1, ex)
    al = ex / den[t - 1]
    pos = (z > 0)
    P = np.zeros((N, H)); np.add.at(P, t - 1, al * pos)
    OP = np.zeros((N, H, C)); np.add.at(OP, t - 1, (al * pos)[:, :, None] * w[s - 1])
    assert np.abs(pp.cpu().numpy() - P).max() <= 2e-6
    assert np.abs(op.cpu().numpy().reshape(N, H, C) - OP).max() <= 1e-5 * max(1.0, np.abs(OP).max())
    # pullback: one edge pass (grad2) against two (grad)
    dy = torch.randn((N, D), generator=gen).cuda()
    dz = dy * (out1 > 0) if act == "relu" else dy
    res = []
    for new in (False, True):
        line, dsd, dss = torch.empty((N, H, 4), **f32), torch.empty((N, H), **f32), torch.empty((N, H), **f32)
        dWx, da = torch.empty((N, D), **f32), torch.empty((H, 2 * C), **f32)
        if new:
            L.check(lib.gnnmp_gat_conv_grad2_f32(plan.handle, plan_t.handle, L.ptr(Wx), None, L.ptr(a), 0.2, L.ptr(st1), L.ptr(out1), L.ptr(b),
                                                 L.ptr(op), L.ptr(pp), L.ptr(dz), L.ptr(line), L.ptr(dsd), L.ptr(dss), L.ptr(dWx), None,
                                                 L.ptr(da), H, C, L.stream_ptr()))
        else:
            L.check(lib.gnnmp_gat_conv_grad_f32(plan.handle, plan_t.handle, L.ptr(Wx), None, L.ptr(a), 0.2, L.ptr(st0), L.ptr(dz), L.ptr(line),
                                                L.ptr(dsd), L.ptr(dss), L.ptr(dWx), None, L.ptr(da), H, C, L.stream_ptr()))
        res.append([v.double().cpu().numpy() for v in (line, dsd, dss, dWx, da)])
    for name, v0, v1 in zip(("line", "dsd", "dss", "dWx", "da"), *res):
        fin = np.isfinite(v0)                    # (the statistics line of a node without in-edges carries m = -inf in both)
        assert np.array_equal(fin, np.isfinite(v1)) and np.array_equal(v0[~fin], v1[~fin]), name
        assert name == "line" or fin.all(), name
        v0, v1 = np.where(fin, v0, 0.0), np.where(fin, v1, 0.0)
        scale = np.abs(v0).max() + 1e-30
        assert np.abs(v0 - v1).max() <= 1e-5 * scale, (name, np.abs(v0 - v1).max() / scale)
        assert np.linalg.norm(v0 - v1) <= 1e-5 * np.linalg.norm(v0) + 1e-30, name
    # nodes without an in-edge: zero contributions, not NaN
    assert np.isfinite(res[1][3]).all() and (res[1][1][-5:] == 0).all()
