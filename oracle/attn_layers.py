"""CPU restatement of the other attention layers on the path — TEST INFRASTRUCTURE ONLY
(same rules as oracle.py: only tests/, smoke() and bench.py's cpu_baseline may import this).

Each body follows the reference statement by statement, materialising the same per-edge temporaries, in float32, on
top of the pinned primitives of oracle.py (gather, scatter, softmax_edge_neighbors, propagate):
    gatv2_conv        GNNlib/src/layers/conv.jl:171-214
    gin_conv          GNNlib/src/layers/conv.jl:250-256
    agnn_conv         GNNlib/src/layers/conv.jl:337-352
    transformer_conv  GNNlib/src/layers/conv.jl:553-629 (no edge features, gating, batch norm, feed-forward block)

PIN STATUS: the reference holds no known-answer vectors for these layers (test/layers/conv.jl only checks output sizes
and gradients through `test_layer`), so the layer bodies themselves are "parity unpinned" beyond their pinned
primitives; tests/test_attn_layers.py guards the restatement with an independent dense-adjacency float64 formulation
of each layer (a different algorithm for the same function).
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

f32 = np.float32


def _lrelu(x, slope):
    # NNlib.leakyrelu(x, a) = ifelse(x > 0, float(x), oftf(x, x * a))
    return np.where(x > 0, x, (x * f32(slope)).astype(f32)).astype(f32)


def _sum_dim1(a):
    """Julia `sum(a, dims = 1)` over the fastest (here: last) axis of a float32 array: sequential accumulation"""
    acc = np.zeros(a.shape[:-1], f32)
    for c in range(a.shape[-1]):
        acc = (acc + a[..., c]).astype(f32)
    return acc


def _heads_tail(y, n, H, C, concat, bias, sigma):
    if not concat:
        acc = y[:, 0, :].copy()
        for h in range(1, H):
            acc = (acc + y[:, h, :]).astype(f32)
        y = (acc / f32(H)).astype(f32)
    y = y.reshape(n, -1)
    if bias is not None:
        y = (y + O._f32(bias)[None, :]).astype(f32)
    return O._act(sigma, y)


def gatv2_conv(s, t, n, x, dense_i_weight, dense_i_bias, dense_j_weight, a, bias=None, sigma=None, heads=1, concat=True,
               negative_slope=0.2, add_self_loops_=True, dropout=0.0, seed=0):
    """a: Julia shape (out, heads), numpy [C, H]; dropout > 0: α = dropout(α, p) (conv.jl:191) with the mask of O.dropout_keep(seed)"""
    s, t = O._i64(s), O._i64(t)
    H = heads
    C = dense_i_weight.shape[0] // H
    if add_self_loops_:
        s, t, _ = O.add_self_loops(s, t, n)
    Wxi = O.matmul(dense_i_weight, x)
    if dense_i_bias is not None:
        Wxi = (Wxi + O._f32(dense_i_bias)[None, :]).astype(f32)
    Wxi = Wxi.reshape(n, H, C)
    Wxj = O.matmul(dense_j_weight, x).reshape(n, H, C)
    ei = O.gather(Wxi, t)
    ej = O.gather(Wxj, s)
    Wx = (ei + ej).astype(f32)                                   # gatv2_message :206
    a_hc = O._f32(np.asarray(a).T)                               # [H, C]
    logit = _sum_dim1((a_hc[None] * _lrelu(Wx, negative_slope)).astype(f32))     # :211  [E', H]
    alpha = O.softmax_edge_neighbors(t, n, logit)
    if dropout > 0.0:
        keep = O.dropout_keep(seed, dropout, len(s), H).astype(f32)
        alpha = (alpha * keep * f32(1.0 / (1.0 - f32(dropout)))).astype(f32)
    beta = (alpha[..., None] * ej).astype(f32)
    y = O.scatter(O.SUM, beta.reshape(len(s), H * C), t, n).reshape(n, H, C)
    return _heads_tail(y, n, H, C, concat, bias, sigma)


def agnn_conv(s, t, n, x, beta=1.0, add_self_loops_=True):
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    if add_self_loops_:
        s, t, _ = O.add_self_loops(s, t, n)
    nrm = np.sqrt(_sum_dim1((x * x).astype(f32))).astype(f32)   # sqrt.(sum(x .^ 2, dims = 1))
    xn = (x / nrm[:, None]).astype(f32)
    cos = _sum_dim1((O.gather(xn, t) * O.gather(xn, s)).astype(f32))[:, None]     # xi_dot_xj: sum(xi .* xj, dims = 1)
    alpha = O.softmax_edge_neighbors(t, n, (f32(beta) * cos).astype(f32))
    m = (alpha * O.gather(x, s)).astype(f32)
    return O.scatter(O.SUM, m, t, n)


def transformer_conv(s, t, n, x, W1, b1, W2, b2, W3, b3, W4, b4, heads=1, concat=True, add_self_loops_=False,
                     skip_connection=False):
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    H = heads
    C = W2.shape[0] // H
    if add_self_loops_:
        s, t, _ = O.add_self_loops(s, t, n)

    def lin(W, b):
        y = O.matmul(W, x)
        return y if b is None else (y + O._f32(b)[None, :]).astype(f32)

    W2x, W3x, W4x = (lin(W, b).reshape(n, H, C) for W, b in ((W2, b2), (W3, b3), (W4, b4)))
    sqrt_out = np.sqrt(f32(C))                                                   # Float32(√out)
    uij = (_sum_dim1((O.gather(W3x, t) * O.gather(W4x, s)).astype(f32)) / sqrt_out).astype(f32)   # :609-616
    alpha = O.softmax_edge_neighbors(t, n, uij)
    val = (alpha[..., None] * O.gather(W2x, s)).astype(f32)
    h = O.scatter(O.SUM, val.reshape(len(s), H * C), t, n).reshape(n, H, C)
    if concat:
        h = h.reshape(n, H * C)
    else:
        acc = h[:, 0, :].copy()
        for k in range(1, H):
            acc = (acc + h[:, k, :]).astype(f32)
        h = (acc / f32(H)).astype(f32)
    if W1 is not None:
        h = (h + lin(W1, b1)).astype(f32)
    if skip_connection:
        h = (h + x).astype(f32)
    return h


def gin_conv(s, t, n, x, eps, aggr=O.SUM):
    """the pre-`nn` part: (1 .+ ϵ) .* xi .+ propagate(copy_xj, g, aggr)"""
    x = O._f32(x)
    m = O.propagate(aggr, s, t, n, x)
    return (((f32(1) + f32(eps)) * x).astype(f32) + m).astype(f32)


def gat_conv_edge(s, t, n, x, e, dense_x_weight, dense_e_weight, a, bias=None, sigma=None, heads=1, concat=True,
                  negative_slope=0.2):
    """gat_conv with edge features (conv.jl:112-167, l.dense_e !== nothing, add_self_loops = false):
    Wxx = vcat(Wxi, Wxj, We); aWW = sum(l.a .* Wxx, dims = 1); logα = leakyrelu.(aWW).  a: Julia shape (3C, H)."""
    s, t = O._i64(s), O._i64(t)
    H = heads
    C = dense_x_weight.shape[0] // H
    Wx = O.matmul(dense_x_weight, x).reshape(n, H, C)
    We = O.matmul(dense_e_weight, O._f32(e)).reshape(len(s), H, C)
    Wxi, Wxj = O.gather(Wx, t), O.gather(Wx, s)
    Wxx = np.concatenate([Wxi, Wxj, We], axis=2)                       # [E, H, 3C]
    a_h = O._f32(np.asarray(a).T)                                      # [H, 3C]
    logit = _lrelu(_sum_dim1((a_h[None] * Wxx).astype(f32)), negative_slope)
    alpha = O.softmax_edge_neighbors(t, n, logit)
    beta = (alpha[..., None] * Wxj).astype(f32)
    y = O.scatter(O.SUM, beta.reshape(len(s), H * C), t, n).reshape(n, H, C)
    return _heads_tail(y, n, H, C, concat, bias, sigma)


# ---------------------------------------------------------------------------------------------------------
# independent float64 formulation: dense masked attention on the adjacency matrix (simple graphs only)
# ---------------------------------------------------------------------------------------------------------
def dense_attention_f64(s, t, n, logit_fn, V, H, C, add_self_loops_):
    """out[i, h] = Σ_j softmax_j(logit_fn(i, j)[h]) V[j, h] over the in-neighbours j of i; logit_fn returns [n, n, H]
    (row = target i, column = source j).  Multi-edges are NOT representable here: callers pass simple graphs."""
    A = np.zeros((n, n), bool)
    A[np.asarray(t) - 1, np.asarray(s) - 1] = True
    if add_self_loops_:
        A[np.arange(n), np.arange(n)] = True
    lg = logit_fn()                                               # [n, n, H]
    lg = np.where(A[..., None], lg, -np.inf)
    mx = lg.max(axis=1, keepdims=True)
    mx = np.where(np.isfinite(mx), mx, 0.0)
    p = np.exp(lg - mx)
    den = p.sum(axis=1, keepdims=True)
    alpha = np.divide(p, den, out=np.zeros_like(p), where=den > 0)
    return np.einsum("ijh,jhc->ihc", alpha, np.asarray(V, np.float64).reshape(n, H, C))
