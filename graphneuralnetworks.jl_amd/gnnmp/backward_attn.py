"""Differentiable GATv2Conv and TransformerConv (attention core + root weight + skip connection) — forward AND backward on
the HIP kernels (csrc/gat_fused.hip, csrc/attn_backward.hip), wrapped as torch.autograd.Functions like gnnmp/backward.py.
The forward keeps 8 bytes of softmax statistics per destination and head; the pullback is two edge passes that rebuild α in
registers, then the dense adjoints.  concat = true or false (mean over heads and its pullback as two leaf kernels); no
edge features / dropout (as in the forward)."""
from __future__ import annotations

import torch

from . import _lib as L
from .backward import _act_code, act_grad, dense_grad_w, dense_grad_x, plan_transposed
from .graph import GNNGraph, check_num_nodes
from .layers import dense
from .layers_attn import ATTN_DOT, ATTN_GATV2


def _attn_forward(plan, mode, Q, K, V, a, slope, scale, bias, act, H, C, p_drop=0.0, seed=0):
    N = plan.n_dst
    out = torch.empty((N, H * C), dtype=torch.float32, device=K.device)
    stats = torch.empty((N, H, 2), dtype=torch.float32, device=K.device)
    if p_drop > 0.0:
        L.check(L.load().gnnmp_attn_conv_drop_f32(plan.handle, mode, L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(a), float(slope),
                                                  float(scale), float(p_drop), int(seed), L.ptr(bias), act, L.ptr(out),
                                                  L.ptr(stats), H, C, L.stream_ptr()))
    else:
        L.check(L.load().gnnmp_attn_conv_f32(plan.handle, mode, L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(a), float(slope), float(scale),
                                             L.ptr(bias), act, L.ptr(out), L.ptr(stats), H, C, L.stream_ptr()))
    return out, stats


def _attn_backward(g, loops, mode, Q, K, V, a, slope, scale, stats, dz, H, C, p_drop=0.0, seed=0):
    plan, plan_t = g.plan(loops), plan_transposed(g, loops)
    N = g.num_nodes
    f32 = dict(dtype=torch.float32, device=dz.device)
    line = torch.empty((N, H, 4), **f32)
    dQ = torch.empty((N, H * C), **f32)
    dK = torch.empty((N, H * C), **f32)
    dV = torch.empty((N, H * C), **f32) if mode == ATTN_DOT else None
    dA = torch.empty((N, H * C), **f32) if mode == ATTN_GATV2 else None
    da = torch.empty((H, C), **f32) if mode == ATTN_GATV2 else None
    if p_drop > 0.0:
        L.check(L.load().gnnmp_attn_conv_grad_drop_f32(plan.handle, plan_t.handle, mode, L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(a),
                                                       float(slope), float(scale), float(p_drop), int(seed), L.ptr(stats), L.ptr(dz),
                                                       L.ptr(line), L.ptr(dQ), L.ptr(dK), L.ptr(dV), L.ptr(dA), L.ptr(da), H, C,
                                                       L.stream_ptr()))
    else:
        L.check(L.load().gnnmp_attn_conv_grad_f32(plan.handle, plan_t.handle, mode, L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(a),
                                                  float(slope), float(scale), L.ptr(stats), L.ptr(dz), L.ptr(line), L.ptr(dQ),
                                                  L.ptr(dK), L.ptr(dV), L.ptr(dA), L.ptr(da), H, C, L.stream_ptr()))
    return dQ, dK, dV, da


def _add_(a, b):
    L.check(L.load().gnnmp_add_f32(L.ptr(a), L.ptr(b), L.ptr(a), a.numel(), L.stream_ptr()))
    return a


class _GATv2ConvFn(torch.autograd.Function):
    """gatv2_conv (GNNlib/src/layers/conv.jl:171-214), e === nothing; concat = true or false"""

    @staticmethod
    def forward(ctx, x, Wi, bi, Wj, a, bias, g, sigma, heads, slope, loops, concat=True, p_drop=0.0, seed=0):
        H, C = heads, Wi.shape[0] // heads
        x = x.contiguous()
        Q = dense(x, Wi, bi)
        K = dense(x, Wj)
        a_hc = a.t().contiguous()                                   # (C, H) as Julia stores it -> [H][C]
        # concat = false: heads averaged before bias and σ (conv.jl:196-200): the kernel's fused tail is off
        out, stats = _attn_forward(g.plan(loops), ATTN_GATV2, Q, K, None, a_hc, slope, 1.0, bias if concat else None,
                                   _act_code(sigma) if concat else L.ACT_IDENTITY, H, C, p_drop, seed)
        ctx.p_drop, ctx.seed = float(p_drop), int(seed)
        if not concat:
            y = torch.empty((out.shape[0], C), dtype=torch.float32, device=x.device)
            L.check(L.load().gnnmp_head_mean_f32(L.ptr(out), L.ptr(bias), _act_code(sigma), L.ptr(y), out.shape[0], H, C,
                                                 L.stream_ptr()))
            out = y
        ctx.save_for_backward(x, Wi, Wj, Q, K, a_hc, stats, out)
        ctx.g, ctx.sigma, ctx.H, ctx.C, ctx.slope, ctx.loops, ctx.concat = g, sigma, H, C, slope, loops, concat
        ctx.has_bi, ctx.has_b = bi is not None, bias is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, Wi, Wj, Q, K, a_hc, stats, y = ctx.saved_tensors
        dz = act_grad(dy.contiguous(), y, ctx.sigma)
        db = dense_grad_w(dz, dz, need_w=False)[1] if ctx.has_b else None
        if not ctx.concat:                       # pullback of mean(x, dims = 2): every head receives Δ / H
            dzh = torch.empty((dz.shape[0], ctx.H * ctx.C), dtype=torch.float32, device=dz.device)
            L.check(L.load().gnnmp_head_mean_grad_f32(L.ptr(dz), L.ptr(dzh), dz.shape[0], ctx.H, ctx.C, L.stream_ptr()))
            dz = dzh
        dQ, dK, _, da = _attn_backward(ctx.g, ctx.loops, ATTN_GATV2, Q, K, None, a_hc, ctx.slope, 1.0, stats, dz, ctx.H, ctx.C,
                                       ctx.p_drop, ctx.seed)
        dWi, dbi = dense_grad_w(dQ, x, need_b=ctx.has_bi)
        dWj, _ = dense_grad_w(dK, x, need_b=False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _add_(dense_grad_x(dQ, Wi), dense_grad_x(dK, Wj))
        return dx, dWi, dbi, dWj, da.t(), db, None, None, None, None, None, None, None, None


def gatv2_conv_ad(l, g: GNNGraph, x, seed=None):
    """differentiable GATv2Conv forward: gradients w.r.t. x, dense_i (weight, bias), dense_j weight, a, bias.  l.dropout > 0: the
    attention coefficients are dropped (conv.jl:191) with the mask of `seed` (default: the layer's next one), regenerated in the pullback"""
    check_num_nodes(g, x)
    assert l.dense_e is None, "the HIP adjoint does not cover edge features"
    p_drop = float(getattr(l, "dropout", 0.0))
    if p_drop > 0.0:
        if seed is None:
            seed = l.next_seed()
        l.last_seed = int(seed)
    return _GATv2ConvFn.apply(x, l.dense_i_weight, l.dense_i_bias, l.dense_j_weight, l.a, l.bias, g, l.sigma, l.heads,
                              l.negative_slope, bool(l.add_self_loops), bool(l.concat), p_drop, 0 if seed is None else int(seed))


class _TransformerConvFn(torch.autograd.Function):
    """transformer_conv (conv.jl:553-629): attention core, + W1 x (root weight), + x (skip connection); concat = true"""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, W3, b3, W4, b4, g, heads, sqrt_out, loops, skip, concat=True):
        H, C = heads, W2.shape[0] // heads
        x = x.contiguous()
        V, Q, K = dense(x, W2, b2), dense(x, W3, b3), dense(x, W4, b4)
        h, stats = _attn_forward(g.plan(loops), ATTN_DOT, Q, K, V, None, 0.0, sqrt_out, None, L.ACT_IDENTITY, H, C)
        if not concat:                                              # mean over heads before the root weight (conv.jl:600-603)
            y = torch.empty((h.shape[0], C), dtype=torch.float32, device=x.device)
            L.check(L.load().gnnmp_head_mean_f32(L.ptr(h), None, L.ACT_IDENTITY, L.ptr(y), h.shape[0], H, C, L.stream_ptr()))
            h = y
        ctx.concat = concat
        if W1 is not None:
            _add_(h, dense(x, W1, b1))
        if skip:
            _add_(h, x)
        ctx.save_for_backward(x, Q, K, V, stats, W2, W3, W4, *([W1] if W1 is not None else []))
        ctx.g, ctx.H, ctx.C, ctx.scale, ctx.loops, ctx.skip = g, H, C, sqrt_out, loops, skip
        ctx.has = (W1 is not None, b1 is not None, b2 is not None, b3 is not None, b4 is not None)
        return h

    @staticmethod
    def backward(ctx, dy):
        x, Q, K, V, stats, W2, W3, W4, *rest = ctx.saved_tensors
        W1 = rest[0] if rest else None
        dh = dy.contiguous()
        dha = dh
        if not ctx.concat:
            dha = torch.empty((dh.shape[0], ctx.H * ctx.C), dtype=torch.float32, device=dh.device)
            L.check(L.load().gnnmp_head_mean_grad_f32(L.ptr(dh), L.ptr(dha), dh.shape[0], ctx.H, ctx.C, L.stream_ptr()))
        dQ, dK, dV, _ = _attn_backward(ctx.g, ctx.loops, ATTN_DOT, Q, K, V, None, 0.0, ctx.scale, stats, dha, ctx.H, ctx.C)
        has_w1, hb1, hb2, hb3, hb4 = ctx.has
        dW2, db2 = dense_grad_w(dV, x, need_b=hb2)
        dW3, db3 = dense_grad_w(dQ, x, need_b=hb3)
        dW4, db4 = dense_grad_w(dK, x, need_b=hb4)
        dW1 = db1 = None
        if has_w1:
            dW1, db1 = dense_grad_w(dh, x, need_b=hb1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dense_grad_x(dV, W2)
            _add_(dx, dense_grad_x(dQ, W3))
            _add_(dx, dense_grad_x(dK, W4))
            if has_w1:
                _add_(dx, dense_grad_x(dh, W1))
            if ctx.skip:
                _add_(dx, dh)
        return dx, dW1, db1, dW2, db2, dW3, db3, dW4, db4, None, None, None, None, None, None


class _AGNNConvFn(torch.autograd.Function):
    """agnn_conv (conv.jl:337-352) as the reference writes it — xn = x ./ norm, α = softmax(β (xn_i . xn_j)), Σ α x_j — so
    that the pullback is the dot-product attention pullback on (Q, K, V) = (xn, xn, x) plus the row normalisation's:
      Δx = ΔV + ∇normalise(ΔQ + ΔK),   Δβ = Σ_e Δlogit_e cos_e = (1 / β) Σ_i xn_i . ΔQ_i   (ΔQ_i = β Σ_e Δlogit_e xn_j)"""

    @staticmethod
    def forward(ctx, x, beta, g, loops):
        x = x.contiguous()
        N, D = x.shape
        b = float(beta)
        assert b != 0.0, "β = 0 makes every logit zero: Δβ is not recoverable from ΔQ (use a non-zero init_beta)"
        lib = L.load()
        xn = torch.empty_like(x)
        rn = torch.empty(N, dtype=torch.float32, device=x.device)
        L.check(lib.gnnmp_row_normalize_f32(L.ptr(x), L.ptr(xn), L.ptr(rn), N, D, L.stream_ptr()))
        out, stats = _attn_forward(g.plan(loops), ATTN_DOT, xn, xn, x, None, 0.0, 1.0 / b, None, L.ACT_IDENTITY, 1, D)
        ctx.save_for_backward(x, xn, rn, stats)
        ctx.g, ctx.loops, ctx.b = g, loops, b
        return out

    @staticmethod
    def backward(ctx, dy):
        x, xn, rn, stats = ctx.saved_tensors
        N, D = x.shape
        dQ, dK, dV, _ = _attn_backward(ctx.g, ctx.loops, ATTN_DOT, xn, xn, x, None, 0.0, 1.0 / ctx.b, stats, dy.contiguous(), 1, D)
        dx = torch.empty_like(x)
        qdot = torch.empty((N, 1), dtype=torch.float32, device=x.device)
        L.check(L.load().gnnmp_row_normalize_grad_f32(L.ptr(dQ), L.ptr(dK), L.ptr(xn), L.ptr(rn), L.ptr(dV), L.ptr(dx),
                                                      L.ptr(qdot), 1.0 / ctx.b, N, D, L.stream_ptr()))
        dbeta = None
        if ctx.needs_input_grad[1]:
            dbeta = dense_grad_w(qdot, qdot, need_w=False)[1]                   # deterministic column sum of (N, 1)
        return dx, dbeta, None, None


def agnn_conv_ad(l, g: GNNGraph, x):
    """differentiable AGNNConv forward: gradients w.r.t. x and β (when `l.beta` is a 1-element tensor that requires grad)"""
    check_num_nodes(g, x)
    beta = l.beta if torch.is_tensor(l.beta) else torch.tensor([float(l.beta)], dtype=torch.float32, device=x.device)
    return _AGNNConvFn.apply(x, beta, g, bool(l.add_self_loops))


def transformer_conv_ad(l, g: GNNGraph, x):
    """differentiable TransformerConv forward (concat = true or false): gradients w.r.t. x and W1..W4 (+ their biases)"""
    check_num_nodes(g, x)
    return _TransformerConvFn.apply(x, l.W1_weight, l.W1_bias, l.W2_weight, l.W2_bias, l.W3_weight, l.W3_bias, l.W4_weight,
                                    l.W4_bias, g, l.heads, l.sqrt_out, bool(l.add_self_loops), bool(l.skip_connection),
                                    bool(l.concat))
