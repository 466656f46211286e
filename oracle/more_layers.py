"""CPU restatement of CGConv, EdgeConv, GatedGraphConv, DConv, NNConv, MEGNetConv, GMMConv, EGNNConv, ChebConv and the
Set2Set pool (GNNlib/src/layers/conv.jl cg_conv :304-333, edge_conv :237-246, gated_graph_conv :218-233, d_conv :696-725,
nn_conv :260-273, megnet_conv :356-368, gmm_conv :372-401, egnn_conv :459-495, cheb_conv :83-98; layers/pool.jl
set2set_pool :31-44) — TEST INFRASTRUCTURE ONLY (same rules as oracle.py).  Statement by
statement on the pinned primitives (gather, scatter, propagate, degree, matmul), float32, in the reference's order: the
per-edge `vcat`s and the (2nin + ein, E) / (2D, E) arrays ARE materialised here, as the reference does.

No known-answer vectors exist in the reference for these layers (test/layers/conv.jl checks sizes and gradients) and the
GRU / LSTM cells live in un-vendored Flux 0.16 (restated from their published definitions: Wi, Wh without bias, one bias
vector b, gates in the order r, z, candidate / input, forget, cell, output; sigmoid_fast / tanh_fast are restated as the
exact functions), ChebConv's eigmax in un-vendored KrylovKit (LAPACK on the same symmetric matrix here) — parity unpinned beyond
the float64 identities tests/test_more_layers.py checks (dense-adjacency formulations)."""
from __future__ import annotations

import numpy as np

from . import oracle as O

f32 = np.float32


def sigmoid(x):
    x = np.asarray(x, f32)
    t = np.exp(-np.abs(x)).astype(f32)
    return np.where(x >= 0, f32(1) / (f32(1) + t), t / (f32(1) + t)).astype(f32)


def softplus(x):
    x = np.asarray(x, f32)
    return (np.log1p(np.exp(-np.abs(x)).astype(f32)).astype(f32) + np.maximum(x, f32(0))).astype(f32)


ACT = {None: lambda v: v, "identity": lambda v: v, "relu": lambda v: np.where(v < 0, f32(0), v).astype(f32),
       "softplus": softplus, "tanh": lambda v: np.tanh(v).astype(f32), "sigmoid": sigmoid}


def _dense(z, W, b, act, blas=True):
    y = O.matmul(O._f32(W), O._f32(z), blas)
    if b is not None:
        y = (y + O._f32(b)[None, :]).astype(f32)
    return ACT[act](y)


def cg_conv(s, t, n, x, e, Wf, bf, Ws, bs, act=None, residual=False):
    """m = propagate(cg_message, g, +): dense_f(z) .* dense_s(z), z = vcat(xi, xj, e); dense_f's σ is sigmoid"""
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    xi, xj = O.gather(x, t), O.gather(x, s)
    z = np.concatenate([xi, xj] + ([] if e is None else [O._f32(e)]), axis=1)
    msg = (_dense(z, Wf, bf, "sigmoid") * _dense(z, Ws, bs, act)).astype(f32)
    m = O.scatter(O.SUM, msg, t, n)
    if residual and m.shape[1] == x.shape[1]:
        m = (m + x).astype(f32)
    return m


def edge_conv(s, t, n, x, nn, aggr="max"):
    """propagate(edge_conv_message, g, aggr): l.nn(vcat(xi, xj .- xi)); nn = [(W, b, act), ...] a chain of Dense layers"""
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    xi, xj = O.gather(x, t), O.gather(x, s)
    z = np.concatenate([xi, (xj - xi).astype(f32)], axis=1)
    for W, b, act in nn:
        z = _dense(z, W, b, act)
    return O.scatter({"max": O.MAX, "+": O.SUM, "mean": O.MEAN, "min": O.MIN}[aggr], z, t, n)


def gru_cell(m, h, Wi, Wh, b):
    """Flux.GRUCell(x = m, h): gxs = chunk(Wi x, 3), ghs = chunk(Wh h, 3), bs = chunk(b, 3)"""
    D = h.shape[1]
    gx, gh = O.matmul(O._f32(Wi), m, True), O.matmul(O._f32(Wh), h, True)
    bb = np.zeros(3 * D, f32) if b is None else O._f32(b)
    r = sigmoid(((gx[:, :D] + gh[:, :D]).astype(f32) + bb[None, :D]).astype(f32))
    z = sigmoid(((gx[:, D:2 * D] + gh[:, D:2 * D]).astype(f32) + bb[None, D:2 * D]).astype(f32))
    c = np.tanh((((gx[:, 2 * D:] + (r * gh[:, 2 * D:]).astype(f32)).astype(f32)) + bb[None, 2 * D:]).astype(f32)).astype(f32)
    return (((f32(1) - z) * c).astype(f32) + (z * h).astype(f32)).astype(f32)


def gated_graph_conv(s, t, n, x, weight, Wi, Wh, b, aggr="+"):
    """weight: [num_layers][dims][dims] (Julia (dims, dims, num_layers) read layer by layer)"""
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    dims = weight.shape[1]
    assert x.shape[1] <= dims
    if x.shape[1] < dims:
        x = np.concatenate([x, np.zeros((n, dims - x.shape[1]), f32)], axis=1)
    h = x
    op = {"+": O.SUM, "mean": O.MEAN, "max": O.MAX, "min": O.MIN}[aggr]
    for i in range(weight.shape[0]):
        m = O.matmul(O._f32(weight[i]), h, True)
        m = O.propagate(op, s, t, n, m, None)
        h = gru_cell(m, h, Wi, Wh, b)
    return h


def d_conv(s, t, n, x, weights, bias, k, edge_weight=None):
    """weights: [2][k][out][in] (Julia (2, k, out, in)); degree(g) is the weighted degree when g has edge weights"""
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    w = None if edge_weight is None else O._f32(edge_weight)
    deg_out, deg_in = O.degree(s, n, w), O.degree(t, n, w)
    W = lambda a, i: O._f32(weights[a][i])
    mm = lambda Wm, v: O.matmul(Wm, v, True)
    fwd = lambda v: O.propagate(O.SUM, s, t, n, O.scale_rows(v, deg_out), w)     # propagate(w_mul_xj, g,  +; xj = T * deg_out')
    bwd = lambda v: O.propagate(O.SUM, t, s, n, O.scale_rows(v, deg_in), w)      # propagate(w_mul_xj, gt, +; xj = T * deg_in)
    h = (mm(W(0, 0), x) + mm(W(1, 0), x)).astype(f32)
    T0 = x
    T1_in = T1_out = None
    if k > 1:
        T1_out, T1_in = fwd(T0), bwd(T0)
        h = ((h + mm(W(0, 1), T1_in)).astype(f32) + mm(W(1, 1), T1_out)).astype(f32)
    for i in range(1, k):       # Julia `for i in 2:l.k` reads l.weights[:, i, :, :] 1-based: slice 2 is used twice (as written)
        T2_in = ((f32(2) * bwd(T1_in)).astype(f32) - T0).astype(f32)
        T2_out = ((f32(2) * fwd(T1_out)).astype(f32) - T0).astype(f32)
        h = ((h + mm(W(0, i), T2_in)).astype(f32) + mm(W(1, i), T2_out)).astype(f32)
        T1_in, T1_out = T2_in, T2_out
    return h if bias is None else (h + O._f32(bias)[None, :]).astype(f32)


def nn_conv(s, t, n, x, e, nn, weight, bias, act=None, aggr="+"):
    """nn_conv (conv.jl:260-273): W_k = reshape(nn(e)[:, k], out, in) per edge (column-major), message W_k x_j, then
    σ.(weight * x .+ m .+ bias).  nn = [(W, b, act), ...] a chain of Dense layers on the edge features."""
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    z = O._f32(e)
    for W, b, a_ in nn:
        z = _dense(z, W, b, a_)
    out, nin = weight.shape
    Wk = z.reshape(len(s), nin, out).transpose(0, 2, 1)             # [E][out][in]: element (o, c) at o + out * c
    xj = O.gather(x, s)
    msg = np.zeros((len(s), out), f32)
    for c in range(nin):                                            # batched_mul: products rounded, summed over in-channels
        msg = (msg + (Wk[:, :, c] * xj[:, c:c + 1]).astype(f32)).astype(f32)
    m = O.scatter({"+": O.SUM, "mean": O.MEAN, "max": O.MAX, "min": O.MIN}[aggr], msg, t, n)
    y = ((O.matmul(O._f32(weight), x, True) + m).astype(f32))
    if bias is not None:
        y = (y + O._f32(bias)[None, :]).astype(f32)
    return ACT[act](y)


def megnet_conv(s, t, n, x, e, phi_e, phi_v, aggr="mean"):
    """megnet_conv (conv.jl:356-368): ē = ϕe(vcat(xi, xj, e)) per edge, xᵉ = aggregate_neighbors(g, aggr, ē),
    x̄ = ϕv(vcat(x, xᵉ)); returns (x̄, ē).  phi_e / phi_v = [(W, b, act), ...] chains of Dense layers."""
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    z = np.concatenate([O.gather(x, t), O.gather(x, s), O._f32(e)], axis=1)
    for W, b, a_ in phi_e:
        z = _dense(z, W, b, a_)
    xe = O.scatter({"+": O.SUM, "mean": O.MEAN, "max": O.MAX, "min": O.MIN}[aggr], z, t, n)
    v = np.concatenate([x, xe], axis=1)
    for W, b, a_ in phi_v:
        v = _dense(v, W, b, a_)
    return v, z


def gmm_conv(s, t, n, x, e, mu, sigma_inv, dense_x_weight, bias, act=None, K=1, residual=False):
    """gmm_conv (conv.jl:372-401).  mu, sigma_inv: [K][ein] (Julia (ein, K)); dense_x_weight: [out * K][in]; the mixture
    weights are exp(+Σ ...) exactly as the reference writes them."""
    s, t = O._i64(s), O._i64(t)
    x, e = O._f32(x), O._f32(e)
    mu, sinv = O._f32(mu), O._f32(sigma_inv)
    out = dense_x_weight.shape[0] // K
    w = ((e[:, None, :] - mu[None, :, :]) ** 2).astype(f32) / f32(2)                # [E][K][ein]
    w = (w * (sinv ** 2).astype(f32)[None]).astype(f32)
    acc = np.zeros(w.shape[:2], f32)
    for d in range(w.shape[2]):
        acc = (acc + w[:, :, d]).astype(f32)
    w = np.exp(acc).astype(f32)                                                     # [E][K]
    xj = O.matmul(O._f32(dense_x_weight), x, True).reshape(n, K, out)               # feature o of kernel k at k * out + o
    msg = (w[:, :, None] * O.gather(xj.reshape(n, K * out), s).reshape(len(s), K, out)).astype(f32)
    m = O.scatter(O.MEAN, msg.reshape(len(s), K * out), t, n).reshape(n, K, out)
    tot = np.zeros((n, out), f32)
    for k in range(K):
        tot = (tot + m[:, k]).astype(f32)
    m = (tot / f32(K)).astype(f32)                                                  # mean(m, dims = 2)
    if bias is not None:
        m = (m + O._f32(bias)[None, :]).astype(f32)
    m = ACT[act](m)
    if residual and x.shape[1] == out:
        m = (m + x).astype(f32)
    return m


def swish(x):
    x = np.asarray(x, f32)
    return (x * sigmoid(x)).astype(f32)


ACT["swish"] = swish


def egnn_conv(s, t, n, h, x, e, phi_e, phi_x, phi_h, residual=False):
    """egnn_conv (conv.jl:459-495): returns (h', x').  phi_* = [(W, b, act), ...] chains of Dense layers."""
    s, t = O._i64(s), O._i64(t)
    h, x = O._f32(h), O._f32(x)
    xd = (O.gather(x, t) - O.gather(x, s)).astype(f32)                       # xi_sub_xj
    sq = np.zeros((len(s), 1), f32)
    for d in range(xd.shape[1]):
        sq = (sq + (xd[:, d:d + 1] * xd[:, d:d + 1]).astype(f32)).astype(f32)
    xd = (xd / (np.sqrt(sq).astype(f32) + f32(1e-6))).astype(f32)
    f = np.concatenate([O.gather(h, t), O.gather(h, s), sq] + ([] if e is None else [O._f32(e)]), axis=1)
    mh = f
    for W, b, a_ in phi_e:
        mh = _dense(mh, W, b, a_)
    mx = mh
    for W, b, a_ in phi_x:
        mx = _dense(mx, W, b, a_)
    mx = (mx * xd).astype(f32)
    h_aggr = O.scatter(O.SUM, mh, t, n)
    x_aggr = O.scatter(O.MEAN, mx, t, n)
    hn = np.concatenate([h, h_aggr], axis=1)
    for W, b, a_ in phi_h:
        hn = _dense(hn, W, b, a_)
    hout = (h + hn).astype(f32) if residual else hn
    return hout, (x + x_aggr).astype(f32)


def scaled_laplacian_dense(s, t, n, edge_weight=None):
    """scaled_laplacian(g, Float32) (GNNGraphs/src/query.jl:442-479) as a dense float64 matrix: A[s, t] (dir = :out, multi-edges
    summed), degs = sum(A, dims = 2), Ã = D^-1/2 A D^-1/2, L = I - Ã, λmax = eigmax(Symmetric(L)), 2/λmax L - I.  The
    reference's eigsolve is KrylovKit (un-vendored); eigmax of the same symmetric matrix by LAPACK here."""
    s, t = O._i64(s), O._i64(t)
    A = np.zeros((n, n))
    np.add.at(A, (s - 1, t - 1), 1.0 if edge_weight is None else np.asarray(edge_weight, np.float64))
    d = A.sum(1)
    assert (d != 0).all(), "Graph contains isolated nodes, cannot compute `normalized_adjacency`."
    c = 1.0 / np.sqrt(d)
    Lm = np.eye(n) - c[:, None] * A * c[None, :]
    lam = np.linalg.eigvalsh(np.triu(Lm) + np.triu(Lm, 1).T)[-1]                   # Symmetric(L): the upper triangle
    return 2.0 / lam * Lm - np.eye(n), lam


def cheb_conv(s, t, n, x, weight, bias, k, edge_weight=None):
    """cheb_conv (conv.jl:83-98): Z_1 = X, Z_2 = X L̃, Z_k = 2 Z_{k-1} L̃ - Z_{k-2}; Y = Σ W_k Z_k + b.  weight: [k][out][in].
    Float32 recurrences on the float32 image of the dense L̃ (the reference holds L̃ as a Float32 matrix)."""
    x = O._f32(x)
    Lt, _ = scaled_laplacian_dense(s, t, n, edge_weight)
    Lt = Lt.astype(f32)
    mmL = lambda Z: (Lt.T.astype(f32) @ Z).astype(f32)                              # (Z * L̃)[:, j] = Σ_i Z[:, i] L̃[i, j]
    Zp, Z = x, mmL(x)
    Y = O.matmul(O._f32(weight[0]), Zp, True)
    if k > 1:
        Y = (Y + O.matmul(O._f32(weight[1]), Z, True)).astype(f32)
    for i in range(2, k):
        Z, Zp = ((f32(2) * mmL(Z)).astype(f32) - Zp).astype(f32), Z
        Y = (Y + O.matmul(O._f32(weight[i]), Z, True)).astype(f32)
    return Y if bias is None else (Y + O._f32(bias)[None, :]).astype(f32)


def set2set_pool(graph_indicator, num_graphs, x, Wi, Wh, b, num_iters):
    """set2set_pool (GNNlib/src/layers/pool.jl:31-44) with Flux.LSTMCell restated (gates input, forget, cell, output of
    Wi x + Wh h + b; un-vendored Flux 0.16).  graph_indicator 1-based; returns [num_graphs][2 n_in]."""
    x = O._f32(x)
    gi = O._i64(graph_indicator)
    n_in = x.shape[1]
    G = num_graphs
    qstar = np.zeros((G, 2 * n_in), f32)
    h = np.zeros((G, n_in), f32)
    c = np.zeros((G, n_in), f32)
    bb = np.zeros(4 * n_in, f32) if b is None else O._f32(b)
    for _ in range(num_iters):
        g4 = ((O.matmul(O._f32(Wi), qstar, True) + O.matmul(O._f32(Wh), h, True)).astype(f32) + bb[None, :]).astype(f32)
        i_, f_, cc, o_ = (g4[:, k * n_in:(k + 1) * n_in] for k in range(4))
        c = ((sigmoid(f_) * c).astype(f32) + (sigmoid(i_) * np.tanh(cc).astype(f32)).astype(f32)).astype(f32)
        h = (sigmoid(o_) * np.tanh(c).astype(f32)).astype(f32)
        q = h
        qn = O.gather(q, gi)                                                      # broadcast_nodes
        sc = np.zeros((x.shape[0], 1), f32)
        for d in range(n_in):
            sc = (sc + (qn[:, d:d + 1] * x[:, d:d + 1]).astype(f32)).astype(f32)
        mx = O.scatter(O.MAX, sc, gi, G)                                          # softmax_nodes
        num = np.exp(sc - O.gather(mx, gi)).astype(f32)
        den = O.scatter(O.SUM, num, gi, G)
        alpha = (num / O.gather(den, gi)).astype(f32)
        r = O.scatter(O.SUM, (x * alpha).astype(f32), gi, G)
        qstar = np.concatenate([q, r], axis=1)
    return qstar
