"""Adjoints of gatv2_conv and transformer_conv restated rule by rule (what Zygote composes from the rrules of gather,
leakyrelu, softmax_edge_neighbors' exp / scatter / division and scatter(+)) — TEST INFRASTRUCTURE ONLY (rules of
oracle.py apply).  float64 arithmetic on the float32 inputs; pinned by central finite differences in
tests/test_backward_attn.py, the same way grad_gat_conv is."""
from __future__ import annotations

import numpy as np

from . import oracle as O


def _softmax_agg(l, ti, n, H):
    m = np.full((n, H), -np.inf)
    np.maximum.at(m, ti, l)
    p = np.exp(l - m[ti])
    den = np.zeros((n, H))
    np.add.at(den, ti, p)
    return p / den[ti]


def _softmax_pullback(alpha, dalpha, ti, n, H):
    sa = np.zeros((n, H))
    np.add.at(sa, ti, alpha * dalpha)
    return alpha * (dalpha - sa[ti])


def grad_gatv2_conv(s, t, n, x, Wi, bi, Wj, a, bias, sigma, dy, heads=1, negative_slope=0.2, add_self_loops_=True,
                    concat=True, dropout=0.0, seed=0):
    """(Δx, ΔWi, Δbi, ΔWj, Δa, Δb); a: Julia shape (C, H)"""
    s, t = O._i64(s), O._i64(t)
    if add_self_loops_:
        s, t, _ = O.add_self_loops(s, t, n)
    si, ti = s - 1, t - 1
    H = heads
    Wi64, Wj64, x64 = (np.asarray(v, np.float64) for v in (Wi, Wj, x))
    C = Wi64.shape[0] // H
    a_hc = np.asarray(a, np.float64).T                              # [H, C]
    Q = (x64 @ Wi64.T + (0 if bi is None else np.asarray(bi, np.float64))).reshape(n, H, C)
    K = (x64 @ Wj64.T).reshape(n, H, C)
    z = Q[ti] + K[si]                                               # [E', H, C]
    lr = np.where(z > 0, z, negative_slope * z)
    l = (a_hc[None] * lr).sum(-1)
    alpha = _softmax_agg(l, ti, n, H)
    # α' = k .* α with the constant k = keep / (1 - p) (conv.jl:191): its rule hands Δα = k .* Δα' to the softmax pullback
    k = O.dropout_keep(seed, dropout, len(ti), H).astype(np.float64) / (1.0 - float(np.float32(dropout))) if dropout > 0.0 else 1.0
    o = np.zeros((n, H, C))
    np.add.at(o, ti, (alpha * k)[..., None] * K[si])
    y = (o.reshape(n, H * C) if concat else o.mean(axis=1)) + (0 if bias is None else np.asarray(bias, np.float64)[None, :])
    dz = np.asarray(dy, np.float64) * (y > 0) if sigma == "relu" else np.asarray(dy, np.float64)
    db = dz.sum(0)
    dzh = dz.reshape(n, H, C) if concat else np.repeat(dz[:, None, :] / H, H, axis=1)   # ∇mean(x, dims = 2)
    delta = dzh[ti]                                                 # Δβ = Δ[t]
    dalpha = (delta * K[si]).sum(-1) * k
    dKj = (alpha * k)[..., None] * delta
    dl = _softmax_pullback(alpha, dalpha, ti, n, H)
    dlr = dl[..., None] * a_hc[None]                                # through sum(a .* lrelu)
    da = (dl[..., None] * lr).sum(0)                                # [H, C]
    dzz = dlr * np.where(z > 0, 1.0, negative_slope)
    dQ = np.zeros((n, H, C))
    dK = np.zeros((n, H, C))
    np.add.at(dQ, ti, dzz)
    np.add.at(dK, si, dzz + dKj)
    dQ, dK = dQ.reshape(n, H * C), dK.reshape(n, H * C)
    f = np.float32
    return ((dQ @ Wi64 + dK @ Wj64).astype(f), (dQ.T @ x64).astype(f), dQ.sum(0).astype(f), (dK.T @ x64).astype(f),
            da.T.astype(f), db.astype(f))


def grad_agnn_conv(s, t, n, x, beta, dy, add_self_loops_=True):
    """(Δx, Δβ) of agnn_conv (GNNlib/src/layers/conv.jl:337-352): xn = x ./ norm, α = softmax(β cos), out = Σ α x_j"""
    s, t = O._i64(s), O._i64(t)
    if add_self_loops_:
        s, t, _ = O.add_self_loops(s, t, n)
    si, ti = s - 1, t - 1
    x64 = np.asarray(x, np.float64)
    b = float(beta)
    r = np.sqrt((x64 * x64).sum(1))
    xn = x64 / r[:, None]
    cos = (xn[ti] * xn[si]).sum(1)[:, None]                          # [E', 1]
    alpha = _softmax_agg(b * cos, ti, n, 1)
    delta = np.asarray(dy, np.float64)[ti]
    dalpha = (delta * x64[si]).sum(1)[:, None]
    dx = np.zeros_like(x64)
    np.add.at(dx, si, alpha * delta)                                # through the message α .* xj
    dl = _softmax_pullback(alpha, dalpha, ti, n, 1)
    dbeta = float((dl * cos).sum())
    dcos = dl * b
    dxn = np.zeros_like(x64)
    np.add.at(dxn, ti, dcos * xn[si])
    np.add.at(dxn, si, dcos * xn[ti])
    dx += (dxn - xn * (xn * dxn).sum(1)[:, None]) / r[:, None]      # through x ./ sqrt.(sum(x .^ 2, dims = 1))
    return dx.astype(np.float32), np.float32(dbeta)


def grad_transformer_conv(s, t, n, x, W1, b1, W2, b2, W3, b3, W4, b4, dy, heads=1, add_self_loops_=False,
                          skip_connection=False, concat=True):
    """(Δx, {name: ΔW / Δb}) for the configuration transformer_conv of oracle/attn_layers.py covers"""
    s, t = O._i64(s), O._i64(t)
    if add_self_loops_:
        s, t, _ = O.add_self_loops(s, t, n)
    si, ti = s - 1, t - 1
    H = heads
    x64 = np.asarray(x, np.float64)
    C = W2.shape[0] // H
    lin = lambda W, b: x64 @ np.asarray(W, np.float64).T + (0 if b is None else np.asarray(b, np.float64))
    V, Q, K = (lin(W, b).reshape(n, H, C) for W, b in ((W2, b2), (W3, b3), (W4, b4)))
    sc = np.sqrt(np.float32(C)).astype(np.float64)
    l = (Q[ti] * K[si]).sum(-1) / sc
    alpha = _softmax_agg(l, ti, n, H)
    dh = np.asarray(dy, np.float64)
    delta = (dh.reshape(n, H, C) if concat else np.repeat(dh[:, None, :] / H, H, axis=1))[ti]   # ∇mean(x, dims = 2)
    dalpha = (delta * V[si]).sum(-1)
    dV = np.zeros((n, H, C))
    np.add.at(dV, si, alpha[..., None] * delta)
    dl = _softmax_pullback(alpha, dalpha, ti, n, H) / sc
    dQ = np.zeros((n, H, C))
    dK = np.zeros((n, H, C))
    np.add.at(dQ, ti, dl[..., None] * K[si])
    np.add.at(dK, si, dl[..., None] * Q[ti])
    dQ, dK, dV = (v.reshape(n, H * C) for v in (dQ, dK, dV))
    g64 = lambda W: np.asarray(W, np.float64)
    dx = dV @ g64(W2) + dQ @ g64(W3) + dK @ g64(W4)
    out = {"W2": dV.T @ x64, "b2": dV.sum(0), "W3": dQ.T @ x64, "b3": dQ.sum(0), "W4": dK.T @ x64, "b4": dK.sum(0)}
    if W1 is not None:
        dx = dx + dh @ g64(W1)
        out["W1"], out["b1"] = dh.T @ x64, dh.sum(0)
    if skip_connection:
        dx = dx + dh
    return dx.astype(np.float32), {k: v.astype(np.float32) for k, v in out.items()}
