"""CPU restatement of SGConv / TAGConv (GNNlib/src/layers/conv.jl sg_conv :501-542, tag_conv :634-685) — TEST
INFRASTRUCTURE ONLY (same rules as oracle.py).  Statement by statement on the pinned primitives (add_self_loops, degree,
scale_rows, propagate, matmul), float32.  No known-answer vectors exist in the reference for these layers
(test/layers/conv.jl checks sizes and gradients): tests/test_khop_layers.py guards the restatement with the float64
dense-matrix identity  sg_conv = W (D^-1/2 (A + I) D^-1/2)^k X  it must satisfy.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

f32 = np.float32


def _prep(s, t, n, edge_weight, add_self_loops_):
    s, t = O._i64(s), O._i64(t)
    w = None if edge_weight is None else O._f32(edge_weight)
    if add_self_loops_:
        s, t, w = O.add_self_loops(s, t, n, w)          # appends ones to the weights (conv.jl:512-515)
    d = O.degree(t, n, w)
    return s, t, w, O.inv_sqrt(d)


def _hop(s, t, n, x, w, c):
    x = O.scale_rows(x, c)                              # x .* c'
    x = O.propagate(O.SUM, s, t, n, x, w)               # e_mul_xj / w_mul_xj / copy_xj, +
    return O.scale_rows(x, c)


def sg_conv(s, t, n, x, weight, bias=None, k=1, add_self_loops_=True, edge_weight=None, blas=True):
    s, t, w, c = _prep(s, t, n, edge_weight, add_self_loops_)
    x = O._f32(x)
    Dout, Din = weight.shape
    if Dout < Din:
        x = O.matmul(weight, x, blas)
    for _ in range(k):
        x = _hop(s, t, n, x, w, c)
    if Dout >= Din:
        x = O.matmul(weight, x, blas)
    return x if bias is None else (x + O._f32(bias)[None, :]).astype(f32)


def tag_conv(s, t, n, x, weight, bias=None, k=3, add_self_loops_=True, edge_weight=None, blas=True):
    s, t, w, c = _prep(s, t, n, edge_weight, add_self_loops_)
    x = O._f32(x)
    sum_pow = sum_total = None
    for it in range(k):
        x = _hop(s, t, n, x, w, c)
        if it == 0:
            sum_pow = x
            sum_total = O.matmul(weight, sum_pow, blas)
        else:
            sum_pow = (sum_pow + x).astype(f32)
            sum_total = (sum_total + O.matmul(weight, sum_pow, blas)).astype(f32)
    return sum_total if bias is None else (sum_total + O._f32(bias)[None, :]).astype(f32)


def _nn_sigmoid(x):
    """NNlib.sigmoid: t = exp(-abs(x)); ifelse(x >= 0, inv(1 + t), t / (1 + t)), float32"""
    x = O._f32(x)
    t = np.exp(-np.abs(x)).astype(f32)
    return np.where(x >= 0, (f32(1) / (f32(1) + t)).astype(f32), (t / (f32(1) + t)).astype(f32)).astype(f32)


def res_gated_graph_conv(s, t, n, x, A, B, U, V, bias=None, sigma=None, blas=True):
    """GNNlib/src/layers/conv.jl:287-300 through the generic path: gather Ax by t, (Bx, Vx) by s, the (D, E) gated
    message, scatter(+), then σ.(U*xi .+ m .+ bias)"""
    s, t = O._i64(s), O._i64(t)
    x = O._f32(x)
    Ax, Bx, Vx, Ux = (O.matmul(W, x, blas) for W in (A, B, V, U))
    msg = (_nn_sigmoid((O.gather(Ax, t) + O.gather(Bx, s)).astype(f32)) * O.gather(Vx, s)).astype(f32)
    m = O.scatter(O.SUM, msg, t, n)
    y = (Ux + m).astype(f32)
    if bias is not None:
        y = (y + O._f32(bias)[None, :]).astype(f32)
    return O._act(sigma, y)
