"""Adjoints of the fused propagate (SURVEY.md §8f rank 1 — the first "next" row after the forward path).

The reference trains through Zygote + NNlib's rrules (∇gather = scatter(+), ∇scatter(+) = gather, ∇scatter(mean) =
gather ./ count, ∇scatter(max|min) = (src .== gather(dst)) .* gather(Δ)) and, on the CPU fast path, the
`adjacency_matrix` rrule (GNNGraphs/src/query.jl:244-278).  Here:

  Δxj of propagate(copy_xj | w_mul_xj | e_mul_xj, g, + | mean)   = the SAME fused kernel on the transposed plan
  Δxj of propagate(copy_xj, g, max | min)                        = gnnmp_propagate_maxmin_grad_f32 (transposed plan)
  Δw  of w_mul_xj / e_mul_xj                                     = gnnmp_edge_dot_f32 (one dot product per edge)

`propagate_ad` wraps them in a torch.autograd.Function so the host mirror can be differentiated end to end (the Julia
extension would declare the same three calls as ChainRulesCore.rrules).
"""
from __future__ import annotations

import torch

from . import _lib as L
from .graph import GNNGraph, Plan, check_num_nodes
from .msgpass import _flat, aggr_code


def plan_transposed(g: GNNGraph, add_self_loops: bool = False) -> Plan:
    """plan of the reversed edge index (t, s): row j lists the edges that LEAVE j, in original edge order"""
    return g.plan_transposed(add_self_loops)


def _in_count(g: GNNGraph, add_self_loops: bool):
    key = ("count", bool(add_self_loops))
    c = g._cache.get(key)
    if c is None:
        c = torch.empty(g.num_nodes, dtype=torch.float32, device=g.device)
        L.check(L.load().gnnmp_degree_f32(g.plan(add_self_loops).handle, None, L.ptr(c), L.stream_ptr()))
        g._cache[key] = c
    return c


def propagate_grad_xj(g: GNNGraph, aggr, dy, w=None, scale_src=None, scale_dst=None, add_self_loops=False, xj=None,
                      y=None):
    """Δxj for out = scale_dst .* aggr_{k: t_k = i}( w_k * scale_src[s_k] * xj[s_k] ).  max/min need the forward input
    `xj` and output `y` (copy_xj only, no scaling — what propagate(copy_xj, g, max) computes)."""
    code = aggr_code(aggr)
    lib = L.load()
    pt = plan_transposed(g, add_self_loops)
    dyf = _flat(dy)
    D = dyf.shape[1]
    dx = torch.empty((g.num_nodes,) + tuple(dy.shape[1:]), dtype=torch.float32, device=dy.device)
    if code in (L.MAX, L.MIN):
        assert w is None and scale_src is None and scale_dst is None and xj is not None and y is not None
        L.check(lib.gnnmp_propagate_maxmin_grad_f32(pt.handle, L.ptr(_flat(xj)), L.ptr(_flat(y)), L.ptr(dyf), L.ptr(dx), D,
                                                    L.stream_ptr()))
        return dx
    sd = scale_dst
    if code == L.MEAN:
        inv = 1.0 / _in_count(g, add_self_loops).clamp(min=1.0)   # N-element host-side plumbing
        sd = inv if sd is None else (sd * inv).contiguous()
    msg = L.COPY_XJ if w is None else L.W_MUL_XJ
    L.check(lib.gnnmp_propagate_f32(pt.handle, msg, L.SUM, L.ptr(dyf), L.ptr(w), L.ptr(sd), L.ptr(scale_src), L.ptr(dx), D,
                                    L.stream_ptr()))
    return dx


def propagate_grad_w(g: GNNGraph, dy, xj, coo_order: bool = False):
    """Δw[k] = Δ[t_k] · xj[s_k]  (w_mul_xj / e_mul_xj with a vector e, aggr = +; for mean pre-scale Δ by 1/count)"""
    dyf, xf = _flat(dy), _flat(xj)
    out = torch.empty(g.num_edges, dtype=torch.float32, device=dy.device)
    D = dyf.shape[1]
    lanes = D // 4 if D % 4 == 0 else (D // 2 if D % 2 == 0 else D)      # the row kernel holds a feature row in ONE lane group (<= 64 lanes)
    if lanes <= 64 and not coo_order:
        # destination-sorted walk: Δ[t] stays in registers for all edges of a destination (half the traffic)
        L.check(L.load().gnnmp_edge_dot_plan_f32(g.plan(False).handle, L.ptr(dyf), L.ptr(xf), L.ptr(out), dyf.shape[1],
                                                 L.stream_ptr()))
        return out
    L.check(L.load().gnnmp_edge_dot_f32(L.ptr(dyf), L.ptr(xf), L.ptr(g.s), L.ptr(g.t), g.idx_bytes, g.index_base,
                                        g.num_edges, dyf.shape[1], L.ptr(out), L.stream_ptr()))
    return out


class _PropagateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xj, w, g, aggr):
        code = aggr_code(aggr)
        lib = L.load()
        plan = g.plan(False)
        xf = _flat(xj)
        out = torch.empty((plan.n_dst, xf.shape[1]), dtype=torch.float32, device=xj.device)
        msg = L.COPY_XJ if w is None else L.W_MUL_XJ
        L.check(lib.gnnmp_propagate_f32(plan.handle, msg, code, L.ptr(xf), L.ptr(w), None, None, L.ptr(out), xf.shape[1],
                                        L.stream_ptr()))
        ctx.g, ctx.aggr, ctx.code = g, aggr, code
        ctx.save_for_backward(xf, w if w is not None else torch.empty(0, device=xj.device), out)
        ctx.has_w = w is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        xf, w, out = ctx.saved_tensors
        g = ctx.g
        w = w if ctx.has_w else None
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if ctx.code in (L.MAX, L.MIN):
                assert w is None, "max/min adjoint is implemented for copy_xj"
                dx = propagate_grad_xj(g, ctx.aggr, dy, xj=xf, y=out)
            else:
                dx = propagate_grad_xj(g, ctx.aggr, dy, w=w)
        if ctx.has_w and ctx.needs_input_grad[1]:
            assert ctx.code in (L.SUM, L.MEAN)
            d = dy
            if ctx.code == L.MEAN:
                d = dy * (1.0 / _in_count(g, False).clamp(min=1.0))[:, None]
            dw = propagate_grad_w(g, d, xf)
        return dx, dw, None, None


def propagate_ad(g: GNNGraph, aggr, xj, w=None):
    """differentiable propagate(copy_xj | w_mul_xj, g, aggr; xj [, w]) — forward and backward both on the HIP kernels"""
    check_num_nodes(g, xj)
    return _PropagateFn.apply(xj, w, g, aggr)


# ---------------------------------------------------------------------------------------------------------
# dense part and the whole GCNConv layer
# ---------------------------------------------------------------------------------------------------------
def _act_code(sigma):
    """fused activation code of an adjoint; ValueError (never an `assert`: those vanish under `python -O`) for anything the HIP
    pullbacks do not cover — treating an unknown σ as identity would hand back silently wrong gradients"""
    from .layers import _act_code as ac
    code, post = ac(sigma)
    if post is not None or code not in (L.ACT_IDENTITY, L.ACT_RELU):
        raise ValueError(f"the HIP adjoints cover σ = identity and relu, not {sigma!r}")
    return code


def act_grad(dy, y, sigma):
    dz = torch.empty_like(dy)
    L.check(L.load().gnnmp_act_grad_f32(L.ptr(dy), L.ptr(y), _act_code(sigma), L.ptr(dz), dy.numel(), L.stream_ptr()))
    return dz


def dense_grad_w(dz, x, need_w=True, need_b=True):
    """(ΔW [Dout, K], Δb [Dout]) = (Δz' * x, colsum(Δz)) — fp32 MFMA, deterministic slab partials"""
    if not need_w and not need_b:
        return None, None
    lib = L.load()
    N, Dout = dz.shape
    K = x.shape[1]
    ws = torch.empty(max(1, lib.gnnmp_dense_grad_workspace(N, Dout, K)), dtype=torch.float32, device=dz.device)
    dW = torch.empty((Dout, K), dtype=torch.float32, device=dz.device) if need_w else None
    db = torch.empty(Dout, dtype=torch.float32, device=dz.device) if need_b else None
    L.check(lib.gnnmp_dense_grad_w_f32(L.ptr(dz), L.ptr(x), N, Dout, K, L.ptr(dW), L.ptr(db), L.ptr(ws), ws.numel(),
                                       L.stream_ptr()))
    return dW, db


def dense_grad_x(dz, W):
    """Δx = Δz * W  ([N, Dout] x [Dout, K]): the forward kernel reading W transposed (w_layout = 1)"""
    N, Dout = dz.shape
    K = W.shape[1]
    assert W.stride(1) == 1 and W.shape[0] == Dout      # a column slice of a wider matrix is fine (sage_conv's halves)
    dx = torch.empty((N, K), dtype=torch.float32, device=dz.device)
    L.check(L.load().gnnmp_dense_f32(L.ptr(dz), L.ptr(W), Dout, W.stride(0), None, None, 0, 0, 1, None, L.ACT_IDENTITY,
                                     L.ptr(dx), N, K, L.stream_ptr()))
    return dx


class _DenseFn(torch.autograd.Function):
    """σ.(W * x .+ b) — Flux.Dense acting on [N, in] rows, forward and backward on the MFMA kernels"""

    @staticmethod
    def forward(ctx, x, weight, bias, sigma):
        from .layers import dense
        x = x.contiguous()
        y = dense(x, weight, bias, sigma)
        ctx.save_for_backward(x, weight, y)
        ctx.sigma, ctx.has_bias = sigma, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dz = act_grad(dy.contiguous(), y, ctx.sigma)
        dW, db = dense_grad_w(dz, x, need_w=ctx.needs_input_grad[1], need_b=ctx.has_bias and ctx.needs_input_grad[2])
        dx = dense_grad_x(dz, weight) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None


def dense_ad(l, x):
    """differentiable gnnmp.Dense forward: gradients w.r.t. x, l.weight, l.bias"""
    return _DenseFn.apply(x, l.weight, l.bias, l.sigma)


class _GCNConvFn(torch.autograd.Function):
    """gcn_conv (GNNlib/src/layers/conv.jl:14-72, default norm, no edge weights) with HIP forward AND backward"""

    @staticmethod
    def forward(ctx, x, weight, bias, g, sigma, add_self_loops):
        from .layers import bias_act, dense, fused_conv, gcn_norm_cache
        lib = L.load()
        plan = g.plan(add_self_loops)
        c, c_slot, _ = gcn_norm_cache(g, add_self_loops)          # degree, 1/sqrt and slot order: once per graph

        def P(h):
            out = torch.empty((plan.n_dst, h.shape[1]), dtype=torch.float32, device=h.device)
            L.check(lib.gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(h), None, L.ptr(c_slot), L.ptr(c), L.ptr(out),
                                                  h.shape[1], L.stream_ptr()))
            return out

        Dout, Din = weight.shape
        x = x.contiguous()
        if Dout < Din:                       # W first (conv.jl:36-40)
            h = dense(x, weight)
            p = P(h)
            y = bias_act(p, bias, sigma)
            ctx.save_for_backward(x, weight, y, c)
        else:
            # the aggregate is the ΔW operand of the backward: the fused kernel hands it out next to the layer output
            r = fused_conv(plan, L.SUM, x, weight, bias, sigma, ss_slot=c_slot, scale_dst=c, return_aggregate=True)
            if r is not None:
                y, agg = r
            else:
                agg = P(x)
                y = dense(agg, weight, bias, sigma)
            ctx.save_for_backward(agg, weight, y, c)
        ctx.g, ctx.sigma, ctx.loops, ctx.w_first, ctx.has_bias = g, sigma, add_self_loops, Dout < Din, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from .layers import fused_conv
        saved, weight, y, c = ctx.saved_tensors
        g = ctx.g
        dz = act_grad(dy.contiguous(), y, ctx.sigma)

        def PT(h):                            # adjoint of the normalised propagate: same kernel, reversed edges
            return propagate_grad_xj(g, "+", h, scale_src=c, scale_dst=c, add_self_loops=ctx.loops)

        if ctx.w_first:
            x = saved
            _, db = dense_grad_w(dz, dz, need_w=False, need_b=ctx.has_bias)
            dh = PT(dz)
            dW, _ = dense_grad_w(dh, x, need_b=False)
            dx = dense_grad_x(dh, weight) if ctx.needs_input_grad[0] else None
        else:
            agg = saved
            dW, db = dense_grad_w(dz, agg, need_b=ctx.has_bias)
            dx = None
            if ctx.needs_input_grad[0]:
                # Δx = PT(Δz * W) = PT(Δz) * W (the propagate acts on rows, W on columns): aggregate-then-transform on the
                # plan of the reversed edges, one kernel, W read as stored
                dx = fused_conv(plan_transposed(g, ctx.loops), L.SUM, dz, weight, scale_src=c, scale_dst=c, w_layout=1)
                if dx is None:
                    dx = PT(dense_grad_x(dz, weight))
        return dx, dW, (db if ctx.has_bias else None), None, None, None


def gcn_conv_ad(l, g: GNNGraph, x):
    """differentiable GCNConv forward (default normalisation, unweighted): gradients w.r.t. x, l.weight, l.bias"""
    check_num_nodes(g, x)
    return _GCNConvFn.apply(x, l.weight, l.bias, g, l.sigma, bool(l.add_self_loops))


# ---------------------------------------------------------------------------------------------------------
# GraphConv / SAGEConv:  y = σ.(W1 * x .+ W2 * propagate(copy_xj, g, aggr; xj = x) .+ b)
# ---------------------------------------------------------------------------------------------------------
class _TwoWeightConvFn(torch.autograd.Function):
    """graph_conv (conv.jl:102-108) and sage_conv (conv.jl:277-283; W = [W1 W2]) with HIP forward AND backward"""

    @staticmethod
    def forward(ctx, x, w1, w2, bias, g, sigma, aggr):
        from .layers import dense
        from .msgpass import _fused
        x = x.contiguous()
        m = _fused(g, L.COPY_XJ, aggr, x, None)
        y = dense(x, w1, bias, sigma, x2=m, W2=w2)
        ctx.save_for_backward(x, m, w1, w2, y)
        ctx.g, ctx.sigma, ctx.aggr, ctx.has_bias = g, sigma, aggr, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, m, w1, w2, y = ctx.saved_tensors
        g = ctx.g
        dz = act_grad(dy.contiguous(), y, ctx.sigma)
        dW1, db = dense_grad_w(dz, x, need_w=ctx.needs_input_grad[1], need_b=ctx.has_bias and ctx.needs_input_grad[3])
        dW2 = dense_grad_w(dz, m, need_b=False)[0] if ctx.needs_input_grad[2] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dm = dense_grad_x(dz, w2)
            if aggr_code(ctx.aggr) in (L.MAX, L.MIN):
                dxm = propagate_grad_xj(g, ctx.aggr, dm, xj=x, y=m)
            else:
                dxm = propagate_grad_xj(g, ctx.aggr, dm)
            dx = dense_grad_x(dz, w1)
            L.check(L.load().gnnmp_add_f32(L.ptr(dx), L.ptr(dxm), L.ptr(dx), dx.numel(), L.stream_ptr()))
        return dx, dW1, dW2, db, None, None, None


def graph_conv_ad(l, g: GNNGraph, x):
    """differentiable GraphConv forward: gradients w.r.t. x, l.weight1, l.weight2, l.bias"""
    check_num_nodes(g, x)
    return _TwoWeightConvFn.apply(x, l.weight1, l.weight2, l.bias, g, l.sigma, l.aggr)


def sage_conv_ad(l, g: GNNGraph, x):
    """differentiable SAGEConv forward: gradients w.r.t. x, l.weight ([W1 W2] column blocks), l.bias"""
    check_num_nodes(g, x)
    Din = x.shape[1]
    return _TwoWeightConvFn.apply(x, l.weight[:, :Din], l.weight[:, Din:], l.bias, g, l.sigma, l.aggr)


# ---------------------------------------------------------------------------------------------------------
# GlobalPool (reduce_nodes over a batch's graph_indicator)
# ---------------------------------------------------------------------------------------------------------
class _GlobalPoolFn(torch.autograd.Function):
    """reduce_nodes(+ | mean, g, x) = scatter(aggr, x, graph_indicator) (GNNlib/src/utils.jl:12-16, layers/pool.jl:3-5) and NNlib's
    rule for it: Δx = gather(Δ, graph_indicator) (./ the member graph's node count for mean) — gnnmp_gather_f32 + gnnmp_mul_rows_f32"""

    @staticmethod
    def forward(ctx, x, g, aggr):
        from .utils import reduce_nodes
        ctx.g, ctx.aggr = g, aggr
        return reduce_nodes(aggr, g, x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        from .graph import graph_indicator
        g = ctx.g
        lib = L.load()
        gi = graph_indicator(g)
        N, G = g.num_nodes, g.num_graphs
        dy = dy.contiguous()
        D = dy.shape[1]
        if ctx.aggr in ("mean",):
            key = "pool_inv_count"
            inv = g._cache.get(key)
            if inv is None:
                # node count of every member graph -> its reciprocal: a constant of the batch (graph prep, cached on the graph)
                from .utils import reduce_nodes
                cnt = reduce_nodes("+", g, torch.ones((N, 1), dtype=torch.float32, device=dy.device)).reshape(G)
                inv = torch.reciprocal(cnt).reshape(G, 1).contiguous()
                g._cache[key] = inv
            scaled = torch.empty_like(dy)
            L.check(lib.gnnmp_mul_rows_f32(L.ptr(inv), 1, L.ptr(dy), L.ptr(scaled), G, D, L.stream_ptr()))
            dy = scaled
        dx = torch.empty((N, D), dtype=torch.float32, device=dy.device)
        L.check(lib.gnnmp_gather_f32(L.ptr(dy), L.ptr(gi), 8 if gi.dtype == torch.int64 else 4, g.index_base, N, L.ptr(dx), D,
                                     L.stream_ptr()))
        return dx, None, None


# ---------------------------------------------------------------------------------------------------------
# The graph-classification chain as ONE differentiable function (round 5)
# ---------------------------------------------------------------------------------------------------------
def dense_grad_w2(dz, x1, x2):
    """(ΔW1, ΔW2, Δb) = (Δz' x1, Δz' x2, colsum(Δz)) from one read of Δz (gnnmp_dense_grad_w2_f32); views of one buffer, each contiguous.
    None when the shape is outside that entry point's envelope (K1 not a multiple of 16)."""
    lib = L.load()
    N, Dout = dz.shape
    K1, K2 = x1.shape[1], x2.shape[1]
    if K1 % 16 != 0:
        return None
    out = torch.empty(Dout * (K1 + K2) + Dout, dtype=torch.float32, device=dz.device)
    ws = torch.empty(max(1, lib.gnnmp_dense_grad_w2_workspace(N, Dout, K1, K2)), dtype=torch.float32, device=dz.device)
    L.check(lib.gnnmp_dense_grad_w2_f32(L.ptr(dz), L.ptr(x1), K1, L.ptr(x2), K2, N, Dout, L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr()))
    return out[: Dout * K1].view(Dout, K1), out[Dout * K1: Dout * (K1 + K2)].view(Dout, K2), out[Dout * (K1 + K2):]


class _GraphChainFn(torch.autograd.Function):
    """GNNChain(GraphConv(σ), ..., GraphConv(σ), GlobalPool(+ | mean), Dense) — examples/graph_classification_tudataset.jl:79-82 — forward
    layer by layer with the activations stored, and a pullback that touches every (N, D) array as few times as the arithmetic allows:
      * Δz_L = relu'(h_L) .* gather(Δpool) ./ count        one pass (gnnmp_pool_grad_act_f32; was mul_rows + gather + act_grad)
      * ΔW_root, ΔW_agg, Δb of a layer                      one read of Δz (gnnmp_dense_grad_w2_f32; was two calls, six fold launches)
      * Δz_{l-1} = relu'(h_{l-1}) .* (Δz_l W_root + Aᵀ(Δz_l W_agg))   the sum and the relu' in the row kernel's epilogue
                                                            (gnnmp_propagate_add_mask_f32; was add + act_grad passes)
    Every value is the one the layer-by-layer composition (graph_conv_ad / global_pool_ad / dense_ad) computes, bit for bit (tested).
    Arguments: x, then (w1, w2, b) per GraphConv, then the head's (W, b)."""

    @staticmethod
    def forward(ctx, x, g, convs, pool_aggr, head_sigma, *params):
        from .layers import dense
        from .msgpass import _fused
        from .utils import reduce_nodes
        x = x.contiguous()
        hs, ms = [x], []
        k = 0
        for (sigma, aggr) in convs:
            w1, w2, b = params[k], params[k + 1], params[k + 2]
            k += 3
            m = _fused(g, L.COPY_XJ, aggr, hs[-1], None)
            hs.append(dense(hs[-1], w1, b, sigma, x2=m, W2=w2))
            ms.append(m)
        pooled = reduce_nodes(pool_aggr, g, hs[-1])
        wh, bh = params[k], params[k + 1]
        y = dense(pooled, wh, bh, head_sigma)
        ctx.save_for_backward(*hs, *ms, pooled, y, *params)
        ctx.g, ctx.convs, ctx.pool_aggr, ctx.head_sigma = g, convs, pool_aggr, head_sigma
        ctx.has_bias = [p is not None for p in params]
        return y

    @staticmethod
    def backward(ctx, dy):
        from .graph import graph_indicator
        g, convs = ctx.g, ctx.convs
        nl = len(convs)
        saved = ctx.saved_tensors
        hs, ms = saved[: nl + 1], saved[nl + 1: 2 * nl + 1]
        pooled, y = saved[2 * nl + 1], saved[2 * nl + 2]
        params = saved[2 * nl + 3:]
        lib = L.load()
        grads = [None] * len(params)
        # head: Dense
        wh, bh = params[3 * nl], params[3 * nl + 1]
        dzh = act_grad(dy.contiguous(), y, ctx.head_sigma)
        grads[3 * nl], grads[3 * nl + 1] = dense_grad_w(dzh, pooled, need_w=True, need_b=bh is not None)
        dpool = dense_grad_x(dzh, wh)
        # pool + relu' of the last GraphConv, one pass
        gi = graph_indicator(g)
        N, G = g.num_nodes, g.num_graphs
        inv = None
        if ctx.pool_aggr == "mean":
            inv = g._cache.get("pool_inv_count")
            if inv is None:
                from .utils import reduce_nodes
                cnt = reduce_nodes("+", g, torch.ones((N, 1), dtype=torch.float32, device=dy.device)).reshape(G)
                inv = g._cache["pool_inv_count"] = torch.reciprocal(cnt).reshape(G, 1).contiguous()
        D = hs[-1].shape[1]
        dz = torch.empty((N, D), dtype=torch.float32, device=dy.device)
        L.check(lib.gnnmp_pool_grad_act_f32(L.ptr(dpool), L.ptr(gi), 8 if gi.dtype == torch.int64 else 4, g.index_base, L.ptr(inv),
                                            L.ptr(hs[-1]), _act_code(convs[-1][0]), L.ptr(dz), N, G, D, L.stream_ptr()))
        for l in range(nl - 1, -1, -1):
            sigma, aggr = convs[l]
            w1, w2, b = params[3 * l], params[3 * l + 1], params[3 * l + 2]
            both = dense_grad_w2(dz, hs[l], ms[l]) if (ctx.needs_input_grad[5 + 3 * l] and ctx.needs_input_grad[6 + 3 * l]) else None
            if both is not None:
                grads[3 * l], grads[3 * l + 1] = both[0], both[1]
                grads[3 * l + 2] = both[2] if b is not None else None
            else:
                grads[3 * l], grads[3 * l + 2] = dense_grad_w(dz, hs[l], need_w=True, need_b=b is not None)
                grads[3 * l + 1] = dense_grad_w(dz, ms[l], need_b=False)[0]
            need_dx = l > 0 or ctx.needs_input_grad[0]
            if not need_dx:
                break
            # Δh_{l} = Δz W_root + Aᵀ(Δz W_agg); for l > 0 times relu'(h_l) of the layer below — all in the row kernel's epilogue
            r = dense_grad_x(dz, w1)
            u = dense_grad_x(dz, w2)
            code = aggr_code(aggr)
            if code not in (L.SUM, L.MEAN):       # (checked before the forward too: graph_chain_ad)
                raise ValueError("the chain's pullback covers + and mean aggregation")
            sd = None
            if code == L.MEAN:
                u_scaled = torch.empty_like(u)      # mean: Δm ./ count BEFORE the transposed sum (propagate_grad_xj's order)
                invc = torch.reciprocal(torch.clamp(_in_count(g, False), min=1.0)).reshape(N, 1).contiguous()
                L.check(lib.gnnmp_mul_rows_f32(L.ptr(invc), 1, L.ptr(u), L.ptr(u_scaled), N, u.shape[1], L.stream_ptr()))
                u = u_scaled
            out = torch.empty_like(r)
            mask = hs[l] if l > 0 else None
            mcode = _act_code(convs[l - 1][0]) if l > 0 else L.ACT_IDENTITY
            L.check(lib.gnnmp_propagate_add_mask_f32(plan_transposed(g, False).handle, L.SUM, L.ptr(u), L.ptr(sd), L.ptr(r),
                                                     L.ptr(mask) if mcode == L.ACT_RELU else None, L.ptr(out), r.shape[1], L.stream_ptr()))
            dz = out
        dx = dz if ctx.needs_input_grad[0] else None
        return (dx, None, None, None, None, *grads)


def graph_chain_ad(model, g: GNNGraph, x):
    """differentiable GNNChain(GraphConv..., GlobalPool, Dense) — the model of BASELINE config 5 — as one autograd function whose
    pullback is scheduled over the whole chain (see _GraphChainFn); gradients w.r.t. x and every layer's parameters"""
    from .layers import Dense, GlobalPool, GraphConv
    layers = model.layers if hasattr(model, "layers") else list(model)
    convs, pool, head = layers[:-2], layers[-2], layers[-1]
    if not (all(isinstance(c, GraphConv) for c in convs) and isinstance(pool, GlobalPool) and isinstance(head, Dense)):
        raise TypeError("graph_chain_ad takes GNNChain(GraphConv..., GlobalPool, Dense)")
    # everything the scheduled pullback cannot differentiate is refused BEFORE the forward runs, with a real exception: a late `assert` in
    # backward (gone under `python -O`) or a dropped relu' mask would mean silently wrong gradients.  Such models train through the
    # layer-by-layer adjoints (graph_conv_ad / global_pool_ad).
    if pool.aggr not in ("+", "sum", "mean"):
        raise ValueError(f"graph_chain_ad: pooling {pool.aggr!r} is outside the chain pullback (+ and mean); use global_pool_ad / the layer adjoints")
    for c in convs:
        if aggr_code(c.aggr) not in (L.SUM, L.MEAN):
            raise ValueError(f"graph_chain_ad: GraphConv aggr {c.aggr!r} is outside the chain pullback (+ and mean); use graph_conv_ad per layer")
        _act_code(c.sigma)                 # ValueError unless identity / relu
    _act_code(head.sigma)
    check_num_nodes(g, x)
    params = []
    for c in convs:
        params += [c.weight1, c.weight2, c.bias]
    params += [head.weight, head.bias]
    return _GraphChainFn.apply(x, g, tuple((c.sigma, c.aggr) for c in convs), "+" if pool.aggr == "sum" else pool.aggr, head.sigma, *params)


def global_pool_ad(l, g: GNNGraph, x):
    """differentiable GlobalPool(+ | mean) over a batched graph"""
    if l.aggr not in ("+", "sum", "mean"):
        raise ValueError(f"global_pool_ad: the pullback covers + and mean pooling, not {l.aggr!r}")
    return _GlobalPoolFn.apply(x, g, "+" if l.aggr == "sum" else l.aggr)


# ---------------------------------------------------------------------------------------------------------
# GATConv
# ---------------------------------------------------------------------------------------------------------
class _GATConvFn(torch.autograd.Function):
    """gat_conv (GNNlib/src/layers/conv.jl:112-167; concat = true, no edge features) with HIP forward AND backward.
    The forward keeps (m, den) per destination and head instead of α; the backward is two edge passes
    (csrc/gat_backward.hip) + the dense adjoints."""

    @staticmethod
    def forward(ctx, x, weight, a, bias, g, sigma, heads, slope, add_self_loops, concat=True, p_drop=0.0, seed=0):
        from .layers import dense
        lib = L.load()
        plan = g.plan(add_self_loops)
        H = heads
        C = weight.shape[0] // H
        N = g.num_nodes
        x = x.contiguous()
        Wx = dense(x, weight)
        a_hc = a.t().contiguous()                                   # [H][2C]
        out = torch.empty((N, H * C), dtype=torch.float32, device=x.device)
        stats = torch.empty((N, H, 2), dtype=torch.float32, device=x.device)
        # concat = false: the heads are averaged BEFORE bias and σ (conv.jl:143-147), so the kernel's fused tail is off
        if p_drop > 0.0:       # α = dropout(α, l.dropout) (conv.jl:139) inside the kernel; the mask is a function of (seed, edge, head)
            L.check(lib.gnnmp_gat_conv_drop_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), float(slope), float(p_drop), int(seed),
                                                L.ptr(bias) if concat else None, _act_code(sigma) if concat else L.ACT_IDENTITY,
                                                L.ptr(out), L.ptr(stats), H, C, L.stream_ptr()))
        else:
            # the training forward also saves o+ / P (csrc/gat_fused.hip ATTN_GAT_PLUS): the pullback's destination side is then a node
            # kernel, not an edge pass (csrc/gat_backward.hip: gat_bwd_node_kernel)
            oplus = torch.empty((N, H * C), dtype=torch.float32, device=x.device)
            pplus = torch.empty((N, H), dtype=torch.float32, device=x.device)
            L.check(lib.gnnmp_gat_conv_train_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), float(slope),
                                                 L.ptr(bias) if concat else None, _act_code(sigma) if concat else L.ACT_IDENTITY,
                                                 L.ptr(out), L.ptr(stats), L.ptr(oplus), L.ptr(pplus), H, C, L.stream_ptr()))
            ctx.plus = (oplus, pplus, out, bias if concat else None)
        if not concat:
            y = torch.empty((N, C), dtype=torch.float32, device=x.device)
            L.check(lib.gnnmp_head_mean_f32(L.ptr(out), L.ptr(bias), _act_code(sigma), L.ptr(y), N, H, C, L.stream_ptr()))
            out = y
        ctx.save_for_backward(x, weight, Wx, a_hc, stats, out)
        ctx.g, ctx.sigma, ctx.H, ctx.C, ctx.slope, ctx.loops, ctx.has_bias = g, sigma, H, C, slope, add_self_loops, bias is not None
        ctx.concat = concat
        ctx.p_drop, ctx.seed = float(p_drop), int(seed)
        if p_drop > 0.0:
            ctx.plus = None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, Wx, a_hc, stats, y = ctx.saved_tensors
        g, H, C = ctx.g, ctx.H, ctx.C
        lib = L.load()
        N = g.num_nodes
        dz = act_grad(dy.contiguous(), y, ctx.sigma)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[3]:
            _, db = dense_grad_w(dz, dz, need_w=False, need_b=True)
        if not ctx.concat:                       # pullback of mean(x, dims = 2): every head receives Δ / H
            dzh = torch.empty((N, H * C), dtype=torch.float32, device=dz.device)
            L.check(lib.gnnmp_head_mean_grad_f32(L.ptr(dz), L.ptr(dzh), N, H, C, L.stream_ptr()))
            dz = dzh
        plan, plan_t = g.plan(ctx.loops), plan_transposed(g, ctx.loops)
        f32 = dict(dtype=torch.float32, device=dz.device)
        line = torch.empty((N, H, 4), **f32)
        dsd = torch.empty((N, H), **f32)
        dss = torch.empty((N, H), **f32)
        dWx = torch.empty((N, H * C), **f32)
        da_hc = torch.empty((H, 2 * C), **f32)
        if ctx.p_drop > 0.0:
            L.check(lib.gnnmp_gat_conv_grad_drop_f32(plan.handle, plan_t.handle, L.ptr(Wx), None, L.ptr(a_hc), float(ctx.slope),
                                                     ctx.p_drop, ctx.seed, L.ptr(stats), L.ptr(dz), L.ptr(line), L.ptr(dsd),
                                                     L.ptr(dss), L.ptr(dWx), None, L.ptr(da_hc), H, C, L.stream_ptr()))
        else:
            oplus, pplus, outk, bk = ctx.plus
            L.check(lib.gnnmp_gat_conv_grad2_f32(plan.handle, plan_t.handle, L.ptr(Wx), None, L.ptr(a_hc), float(ctx.slope),
                                                 L.ptr(stats), L.ptr(outk), L.ptr(bk), L.ptr(oplus), L.ptr(pplus), L.ptr(dz), L.ptr(line),
                                                 L.ptr(dsd), L.ptr(dss), L.ptr(dWx), None, L.ptr(da_hc), H, C, L.stream_ptr()))
        dW = dense_grad_w(dWx, x, need_b=False)[0] if ctx.needs_input_grad[1] else None
        dx = dense_grad_x(dWx, weight) if ctx.needs_input_grad[0] else None
        return dx, dW, da_hc.t(), db, None, None, None, None, None, None, None, None


def gat_conv_ad(l, g: GNNGraph, x, seed=None):
    """differentiable GATConv forward (no edge features; concat = true or false): gradients w.r.t. x, l.dense_x_weight,
    l.a, l.bias.  l.dropout > 0: the attention coefficients are dropped (conv.jl:139) with the mask of `seed` (default: the layer's
    next seed), in the forward and — recomputed, never stored — in the pullback."""
    check_num_nodes(g, x)
    assert getattr(l, "dense_e_weight", None) is None, "the HIP adjoint does not cover edge features"
    p_drop = float(getattr(l, "dropout", 0.0))
    if p_drop > 0.0:
        if seed is None:
            seed = l.next_seed()
        l.last_seed = int(seed)
    return _GATConvFn.apply(x, l.dense_x_weight, l.a, l.bias, g, l.sigma, l.heads, l.negative_slope,
                            bool(l.add_self_loops), bool(l.concat), p_drop, 0 if seed is None else int(seed))
