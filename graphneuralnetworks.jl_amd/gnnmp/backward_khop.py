"""Differentiable SGConv / TAGConv: the GCN-normalised hop and its pullback as one autograd Function (both directions are one
fused kernel launch: the pullback of `c .* (A_w (c .* x))` is the same hop on the plan of (t, s)), composed with the dense
adjoints of gnnmp/backward.py.  Gradients w.r.t. x, weight, bias (edge weights are treated as constants, as the layers'
`use_edge_weight` path does in the reference: the normalisation is `ignore_derivatives`-wrapped there too)."""
from __future__ import annotations

import torch

from . import _lib as L
from .backward import _DenseFn, propagate_grad_xj
from .graph import GNNGraph, check_num_nodes
from .layers_khop import _edge_weights, _hop, _norm_slots


class _HopFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, loops, c, ss, ws, w):
        ctx.g, ctx.loops, ctx.c, ctx.w = g, loops, c, w
        return _hop(g, loops, x.contiguous(), c, ss, ws)

    @staticmethod
    def backward(ctx, dy):
        dx = propagate_grad_xj(ctx.g, "+", dy.contiguous(), w=ctx.w, scale_src=ctx.c, scale_dst=ctx.c, add_self_loops=ctx.loops)
        return dx, None, None, None, None, None, None


class _AddFn(torch.autograd.Function):
    """a + b on the device (gnnmp_add_f32); the pullback hands Δ to both"""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        L.check(L.load().gnnmp_add_f32(L.ptr(a.contiguous()), L.ptr(b.contiguous()), L.ptr(out), a.numel(), L.stream_ptr()))
        return out

    @staticmethod
    def backward(ctx, d):
        return d, d


def sg_conv_ad(l, g: GNNGraph, x, edge_weight=None):
    """differentiable sg_conv (GNNlib/src/layers/conv.jl:501-542)"""
    check_num_nodes(g, x)
    w = _edge_weights(l, g, edge_weight)
    loops = bool(l.add_self_loops)
    c, ss, ws = _norm_slots(g, loops, w)
    Dout, Din = l.weight.shape
    if Dout < Din:
        x = _DenseFn.apply(x, l.weight, None, None)
    for _ in range(l.k):
        x = _HopFn.apply(x, g, loops, c, ss, ws, w)
    if Dout >= Din:
        return _DenseFn.apply(x, l.weight, l.bias, None)
    return _BiasFn.apply(x, l.bias) if l.bias is not None else x


class _BiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b):
        from .layers import bias_act
        return bias_act(x.contiguous(), b, None)

    @staticmethod
    def backward(ctx, d):
        from .backward import dense_grad_w
        d = d.contiguous()
        return d, dense_grad_w(d, d, need_w=False)[1]


def tag_conv_ad(l, g: GNNGraph, x, edge_weight=None):
    """differentiable tag_conv (conv.jl:634-685): Σ_iter W (Σ_{j <= iter} Ã^j x) + b"""
    check_num_nodes(g, x)
    w = _edge_weights(l, g, edge_weight)
    loops = bool(l.add_self_loops)
    c, ss, ws = _norm_slots(g, loops, w)
    sum_pow = total = None
    for it in range(l.k):
        x = _HopFn.apply(x, g, loops, c, ss, ws, w)
        sum_pow = x if it == 0 else _AddFn.apply(sum_pow, x)
        term = _DenseFn.apply(sum_pow, l.weight, None, None)
        total = term if it == 0 else _AddFn.apply(total, term)
    return _BiasFn.apply(total, l.bias) if l.bias is not None else total


class _AxpyFn(torch.autograd.Function):
    """alpha .* x .+ y on the device; alpha a constant (gin_conv's ϵ is not trainable in the reference)"""

    @staticmethod
    def forward(ctx, x, y, alpha):
        ctx.alpha = float(alpha)
        out = torch.empty_like(x)
        L.check(L.load().gnnmp_axpy_f32(ctx.alpha, L.ptr(x.contiguous()), L.ptr(y.contiguous()), L.ptr(out), x.numel(),
                                        L.stream_ptr()))
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        dx = torch.empty_like(d)
        zero = torch.zeros_like(d)
        L.check(L.load().gnnmp_axpy_f32(ctx.alpha, L.ptr(d), L.ptr(zero), L.ptr(dx), d.numel(), L.stream_ptr()))
        return dx, d, None


def gin_conv_ad(l, g: GNNGraph, x):
    """differentiable gin_conv (conv.jl:250-256) for `nn` a gnnmp Dense or a list of them: gradients w.r.t. x and the Dense
    parameters; propagate(copy_xj, aggr) through its own pullback (any aggr)"""
    from .backward import propagate_ad
    check_num_nodes(g, x)
    m = propagate_ad(g, l.aggr, x)
    z = _AxpyFn.apply(x, m, 1.0 + float(l.eps))
    for layer in (l.nn if isinstance(l.nn, (list, tuple)) else [l.nn]):
        z = _DenseFn.apply(z, layer.weight, layer.bias, layer.sigma)
    return z
