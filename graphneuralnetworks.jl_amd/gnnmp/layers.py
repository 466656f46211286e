"""Functional layer bodies and thin layer objects for the four convolutions on the hot path + GlobalPool.

Mirror of GNNlib/src/layers/conv.jl (functional bodies taking a bag of fields `l`) and of the Flux front-end structs in
GraphNeuralNetworks/src/layers/conv.jl (same constructor arguments, field names and defaults):
  gcn_conv      GNNlib conv.jl:14-72     GCNConv     front-end conv.jl:77-104   (add_self_loops=true, use_edge_weight=false)
  graph_conv    GNNlib conv.jl:102-108   GraphConv   front-end conv.jl:226-245  (aggr = +)
  gat_conv      GNNlib conv.jl:112-167   GATConv     front-end conv.jl:309-346  (heads=1, concat=true, slope=0.2)
  sage_conv     GNNlib conv.jl:277-283   SAGEConv    front-end conv.jl:770-787  (aggr = mean)
  global_pool   GNNlib pool.jl:3-5       GlobalPool  front-end pool.jl:35-41

Weights keep the Julia shapes: `weight (out, in)` etc., stored as row-major torch tensors [out, in].
Every arithmetic step is a libgnnmp call (HIP); torch only allocates.
"""
from __future__ import annotations

import math

import numpy as np

import torch

from . import _lib as L
from . import placement
from .graph import GNNGraph, check_num_nodes, degree
from .msgpass import _flat, _fused, aggr_code
from .utils import expand_srcdst, reduce_nodes

_ACT = {None: L.ACT_IDENTITY, "identity": L.ACT_IDENTITY, "relu": L.ACT_RELU, torch.relu: L.ACT_RELU,
        torch.nn.functional.relu: L.ACT_RELU}


def _act_code(sigma):
    """(fused code or None, python callable or None)"""
    if sigma in _ACT:
        return _ACT[sigma], None
    if callable(sigma):
        return L.ACT_IDENTITY, sigma
    raise ValueError(f"unsupported activation {sigma!r}")


def dense(x, W, bias=None, sigma=None, x2=None, W2=None, out=None):
    """act(W * x (+ W2 * x2) (+ bias)) on the MFMA kernel.  W: [Dout, Din] (Julia (out, in)); W/W2 may be column
    slices of a wider matrix (sage_conv's `weight * vcat(xi, m)`): only the last stride must be 1."""
    assert x.dtype == torch.float32 and W.dtype == torch.float32
    x = x.contiguous()
    N, D1 = x.shape
    Dout = W.shape[0]
    assert W.shape[1] == D1 and W.stride(1) == 1
    D2 = 0
    ld2 = 0
    if x2 is not None:
        x2 = x2.contiguous()
        D2 = x2.shape[1]
        assert W2.shape == (Dout, D2) and W2.stride(1) == 1 and x2.shape[0] == N
        ld2 = W2.stride(0)
    code, post = _act_code(sigma)
    if out is None:
        out = torch.empty((N, Dout), dtype=torch.float32, device=x.device)
    else:
        assert out.shape == (N, Dout) and out.dtype == torch.float32 and out.is_contiguous()
    b = None if bias is None or bias is False else bias.contiguous()
    pr = L._probe
    e0 = pr.begin() if pr is not None else None
    L.check(L.load().gnnmp_dense_f32(L.ptr(x), L.ptr(W), D1, W.stride(0), L.ptr(x2), L.ptr(W2), D2, ld2, 0,
                                     L.ptr(b), code, L.ptr(out), N, Dout, L.stream_ptr()))
    if pr is not None:
        pr.end("dense", e0)
    return post(out) if post is not None else out


def fused_conv(plan, aggr, xj, W_agg, bias=None, sigma=None, xi=None, W_root=None, w=None, scale_src=None, w_slot=None,
               ss_slot=None, scale_dst=None, return_aggregate=False, w_layout=0, out=None):
    """act(W_root * xi + W_agg * A + bias) with A = the plan's aggregation of xj (same arguments as gnnmp_propagate_f32 /
    gnnmp_propagate_slots_f32), in ONE kernel: A stays in LDS (csrc/fused_conv.hip).  Returns None when the shape is outside
    the kernel's envelope (the caller then runs propagate + dense); with return_aggregate the pre-GEMM aggregate comes back
    too (bit-identical to the unfused propagate: tests).  w_layout = 1: W_agg is [D, Dout] and the product is A * W_agg (the
    adjoint's `Δz * W` with the layer's own [Dout, Din] matrix, no transposed copy); not with a root term."""
    xf = _flat(xj)
    D = xf.shape[1]
    if w_layout:
        assert xi is None
        Dout = W_agg.shape[1]
        assert W_agg.shape[0] == D and W_agg.stride(1) == 1
    else:
        Dout = W_agg.shape[0]
        assert W_agg.shape[1] == D and W_agg.stride(1) == 1
    D1 = 0
    if xi is not None:
        xi = _flat(xi)
        D1 = xi.shape[1]
        assert W_root.shape == (Dout, D1) and W_root.stride(1) == 1
    code, post = _act_code(sigma)
    b = None if bias is None or bias is False else bias.contiguous()
    lib = L.load()
    # the library's default gating, mirrored here so that the shapes it would refuse anyway cost neither two throw-away allocations nor a
    # failed call with a formatted error string: with knob 14 at its default the kernel fuses only a layer without a root term
    # whose aggregate exceeds 128 MiB (csrc/fused_conv.hip); knob 14 > 0 forces it, < 0 disables it
    k14 = _knob14()
    # (round 6: sage_conv / graph_conv 100 + 100 => a multiple of 128 outputs has a fused kernel too — fused_cat_kernel — but it loses to
    # the two-kernel path on the products shape, 6.85 against 5.77 ms, and runs only when knob 14 > 0 forces it)
    if k14 < 0 or (k14 == 0 and (D1 > 0 or plan.n_dst * D * 4 < (128 << 20))):
        return None
    if out is None:
        out = torch.empty((plan.n_dst, Dout), dtype=torch.float32, device=xf.device)
    else:
        assert out.shape == (plan.n_dst, Dout) and out.dtype == torch.float32 and out.is_contiguous()
    agg = torch.empty((plan.n_dst, D), dtype=torch.float32, device=xf.device) if return_aggregate else None
    pr = L._probe
    e0 = pr.begin() if pr is not None else None
    rc = lib.gnnmp_fused_conv_f32(plan.handle, aggr, L.ptr(xf), L.ptr(w), L.ptr(scale_src), L.ptr(w_slot), L.ptr(ss_slot),
                                  L.ptr(scale_dst), D, L.ptr(xi), D1, L.ptr(W_root), 0 if W_root is None else W_root.stride(0),
                                  L.ptr(W_agg), W_agg.stride(0), int(w_layout), L.ptr(b), code, L.ptr(out), Dout, L.ptr(agg), L.stream_ptr())
    if pr is not None:
        pr.end("fused_conv", e0)
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc)
    out = post(out) if post is not None else out
    return (out, agg) if return_aggregate else out


_KNOB14 = [0]


def _would_fuse(plan, D):
    """the library's default gating of fused_conv for a layer without a root term (see fused_conv above)"""
    k14 = _KNOB14[0]
    return k14 > 0 or (k14 == 0 and plan.n_dst * D * 4 >= (128 << 20))


def _knob14():
    return _KNOB14[0]


def bias_act(x, bias, sigma):
    code, post = _act_code(sigma)
    b = None if bias is None or bias is False else bias.contiguous()
    if b is None and code == L.ACT_IDENTITY:
        return post(x) if post is not None else x
    xf = _flat(x)
    out = torch.empty_like(xf)
    L.check(L.load().gnnmp_bias_act_f32(L.ptr(xf), L.ptr(b), code, L.ptr(out), xf.shape[0], xf.shape[1], L.stream_ptr()))
    out = out.view(x.shape)
    return post(out) if post is not None else out


def _inv_sqrt(d):
    out = torch.empty_like(d)
    L.check(L.load().gnnmp_inv_sqrt_f32(L.ptr(d), L.ptr(out), d.numel(), L.stream_ptr()))
    return out


# ---------------------------------------------------------------------------------------------------------
# GCNConv
# ---------------------------------------------------------------------------------------------------------
def gcn_norm_cache(g: GNNGraph, loops: bool, w=None):
    """(c, c_slot, w_slot) for the default normalisation with the graph's own weights (`w is g.w`) or none: c = 1 ./ sqrt.(degree(g;
    dir = :in)) on the (self-looped) graph (conv.jl:52-56), and the per-edge source coefficient / weight laid out in the plan's
    slot order so that the fused kernel streams them coalesced.  They depend only on the graph: computed on the first call and
    cached on it (the reference recomputes degree and `xj .* cout'` on every call, conv.jl:52-59); bench.py reports the one-off
    cost as `norm_cache_ms`."""
    assert w is None or w is g.w
    key = ("gcn_norm", bool(loops), w is not None)
    hit = g._cache.get(key)
    if hit is not None:
        return hit
    lib = L.load()
    plan = g.plan(loops)
    d = torch.empty(g.num_nodes, dtype=torch.float32, device=g.device)
    L.check(lib.gnnmp_degree_f32(plan.handle, L.ptr(w), L.ptr(d), L.stream_ptr()))
    c = _inv_sqrt(d)
    c_slot = torch.empty(plan.n_total, dtype=torch.float32, device=g.device)
    L.check(lib.gnnmp_plan_slot_gather_f32(plan.handle, 0, L.ptr(c), L.ptr(c_slot), L.stream_ptr()))
    w_slot = None
    if w is not None:
        w_slot = torch.empty(plan.n_total, dtype=torch.float32, device=g.device)
        L.check(lib.gnnmp_plan_slot_gather_f32(plan.handle, 1, L.ptr(w), L.ptr(w_slot), L.stream_ptr()))
    g._cache[key] = (c, c_slot, w_slot)
    return g._cache[key]


def gcn_conv(l, g: GNNGraph, x, edge_weight=None, norm_fn=None, conv_weight=None):
    """GNNlib/src/layers/conv.jl:14-72.  One fused propagate: the two normalisation passes (`xj .* cout'`,
    `x .* cin'`), the self loops and the edge weights all live inside gnnmp_propagate_f32."""
    if edge_weight is not None:
        edge_weight = edge_weight.to(device=g.device, dtype=torch.float32).contiguous()
        if edge_weight.numel() != g.num_edges:
            raise ValueError(f"Wrong number of edge weights (expected {g.num_edges} but given {edge_weight.numel()})")
    weight = l.weight
    if conv_weight is not None:
        if tuple(conv_weight.shape) != tuple(l.weight.shape):
            raise ValueError(f"The weight matrix has the wrong size. Expected {tuple(l.weight.shape)} "
                             f"but got {tuple(conv_weight.shape)}")
        weight = conv_weight
    check_num_nodes(g, x)
    loops = bool(l.add_self_loops)
    plan = g.plan(loops)
    Dout, Din = weight.shape
    if Dout < Din:
        x = dense(x, weight)  # multiply before convolution if it is more convenient (conv.jl:36-40)
    # degree(g, T; dir = :in, edge_weight) on the self-looped graph (conv.jl:52-56)
    if edge_weight is not None:
        w = edge_weight
    elif l.use_edge_weight:
        w = g.w
    else:
        w = None
    lib = L.load()
    cacheable = norm_fn is None and edge_weight is None
    if cacheable:
        c, c_slot, w_slot = gcn_norm_cache(g, loops, w)
    else:
        d = torch.empty(g.num_nodes, dtype=torch.float32, device=g.device)
        L.check(lib.gnnmp_degree_f32(plan.handle, L.ptr(w), L.ptr(d), L.stream_ptr()))
        c = _inv_sqrt(d) if norm_fn is None else norm_fn(d).to(torch.float32).contiguous()
        c_slot = w_slot = None
    if Dout >= Din:
        # aggregate, then transform: one kernel, the (N, Din) aggregate never goes to HBM (conv.jl:59-71)
        # opt-in (gnnmp/placement.py): a persistent output buffer in a placement class other than x's, the best of the candidates by trial
        out_buf = ch = tok = None
        if placement.enabled(l) and x.dim() == 2 and placement.worth_it((plan.n_dst, Dout)) and _would_fuse(plan, x.shape[1]):
            ar = placement.arena()
            if ar is not None:
                ch = placement.choice_for(l, "out", (plan.n_dst, Dout), [ar.class_of(x)])
                if ch is not None:
                    out_buf, tok = ch.begin()
        if c_slot is not None:
            y = fused_conv(plan, L.SUM, x, weight, l.bias, l.sigma, w_slot=w_slot, ss_slot=c_slot, scale_dst=c, out=out_buf)
        else:
            y = fused_conv(plan, L.SUM, x, weight, l.bias, l.sigma, w=w, scale_src=c, scale_dst=c, out=out_buf)
        if ch is not None:
            ch.end(tok)
        if y is not None:
            return y
    if c_slot is not None:
        xf = _flat(x)
        out = torch.empty((plan.n_dst, xf.shape[1]), dtype=torch.float32, device=x.device)
        if Dout < Din:
            # W came first: bias and σ ride in the row kernel's epilogue (conv.jl:71), no separate pass over (N, Dout)
            code, post = _act_code(l.sigma)
            b = None if l.bias is None or l.bias is False else l.bias.contiguous()
            L.check(lib.gnnmp_propagate_slots_act_f32(plan.handle, L.SUM, L.ptr(xf), L.ptr(w_slot), L.ptr(c_slot), L.ptr(c),
                                                      L.ptr(b), code, L.ptr(out), xf.shape[1], L.stream_ptr()))
            return post(out) if post is not None else out
        L.check(lib.gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(xf), L.ptr(w_slot), L.ptr(c_slot), L.ptr(c),
                                              L.ptr(out), xf.shape[1], L.stream_ptr()))
        x = out
    else:
        msg = L.COPY_XJ if w is None else L.W_MUL_XJ
        x = _fused(g, msg, "+", x, w, scale_src=c, scale_dst=c, add_self_loops=loops)
    if Dout >= Din:
        return dense(x, weight, l.bias, l.sigma)
    return bias_act(x, l.bias, l.sigma)


class GCNConv:
    """GCNConv(in => out, σ=identity; bias=true, add_self_loops=true, use_edge_weight=false)
    — GraphNeuralNetworks/src/layers/conv.jl:77-104"""

    def __init__(self, ch, sigma=None, bias=True, add_self_loops=True, use_edge_weight=False, device="cuda", seed=None):
        cin, cout = ch
        self.weight = glorot_uniform(cout, cin, device=device, seed=seed)
        self.bias = torch.zeros(cout, dtype=torch.float32, device=device) if bias else None
        self.sigma = sigma
        self.add_self_loops = add_self_loops
        self.use_edge_weight = use_edge_weight

    def __call__(self, g, x, edge_weight=None, norm_fn=None, conv_weight=None):
        return gcn_conv(self, g, x, edge_weight, norm_fn, conv_weight)


# ---------------------------------------------------------------------------------------------------------
# GraphConv
# ---------------------------------------------------------------------------------------------------------
def graph_conv(l, g: GNNGraph, x):
    """GNNlib/src/layers/conv.jl:102-108: σ.(W1*xi .+ W2*propagate(copy_xj, g, aggr) .+ b)"""
    check_num_nodes(g, x)
    xj, xi = expand_srcdst(g, x)
    y = fused_conv(g.plan(False), aggr_code(l.aggr), xj, l.weight2, l.bias, l.sigma, xi=xi, W_root=l.weight1)
    if y is not None:
        return y
    mb, ch, tok = _placed_aggregate(l, xj, g.plan(False).n_dst)
    m = _fused(g, L.COPY_XJ, l.aggr, xj, None, out=mb)
    if ch is not None:
        ch.end(tok)
    return dense(xi, l.weight1, l.bias, l.sigma, x2=m, W2=l.weight2)


class GraphConv:
    """GraphConv(in => out, σ=identity; aggr=+, bias=true) — GraphNeuralNetworks/src/layers/conv.jl:226-245"""

    def __init__(self, ch, sigma=None, aggr="+", bias=True, device="cuda", seed=None):
        cin, cout = ch
        self.weight1 = glorot_uniform(cout, cin, device=device, seed=seed)
        self.weight2 = glorot_uniform(cout, cin, device=device, seed=None if seed is None else seed + 1)
        self.bias = torch.zeros(cout, dtype=torch.float32, device=device) if bias else None
        self.sigma = sigma
        self.aggr = aggr

    def __call__(self, g, x):
        return graph_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# SAGEConv
# ---------------------------------------------------------------------------------------------------------
def sage_conv(l, g: GNNGraph, x):
    """GNNlib/src/layers/conv.jl:277-283: σ.(W * vcat(xi, propagate(copy_xj, g, aggr)) .+ b); the vcat is never built —
    the first `in` columns of W multiply xi, the last `in` multiply the aggregate."""
    check_num_nodes(g, x)
    xj, xi = expand_srcdst(g, x)
    Din = xi.shape[1]
    W = l.weight
    y = fused_conv(g.plan(False), aggr_code(l.aggr), xj, W[:, Din:], l.bias, l.sigma, xi=xi, W_root=W[:, :Din])
    if y is not None:
        return y
    mb, ch, tok = _placed_aggregate(l, xj, g.plan(False).n_dst)
    m = _fused(g, L.COPY_XJ, l.aggr, xj, None, out=mb)
    if ch is not None:
        ch.end(tok)
    # A second, explicit opt-in (`l.persistent_out = True`): the layer's output too is a persistent arena buffer (a buffer larger than one
    # 2 GiB block — (N, 256) on the products shape — is an allocation of its own, gnnmp_arena_alloc), in a class other than the
    # aggregate's.  Without it only the INTERNAL aggregate is persistent and the returned tensor is an ordinary allocation: a layer
    # applied twice, or outputs collected across iterations, must not alias because placement was switched on globally.
    ob = None
    if mb is not None and getattr(l, "persistent_out", False):
        ar = placement.arena()
        ob, _ = placement.buffer_for(l, "out", (m.shape[0], W.shape[0]), [ar.class_of(m)])
    return dense(xi, W[:, :Din], l.bias, l.sigma, x2=m, W2=W[:, Din:], out=ob)


def _placed_aggregate(l, xj, n_dst):
    """opt-in (gnnmp/placement.py): (the layer's persistent aggregate buffer in a placement class other than xj's, its Choice, token);
    (None, None, None) = allocate as usual"""
    if not (placement.enabled(l) and xj.dim() == 2 and placement.worth_it((n_dst, xj.shape[1]))):
        return None, None, None
    ar = placement.arena()
    if ar is None:
        return None, None, None
    ch = placement.choice_for(l, "m", (n_dst, xj.shape[1]), [ar.class_of(xj)])
    if ch is None:
        return None, None, None
    buf, tok = ch.begin()
    return buf, ch, tok


class SAGEConv:
    """SAGEConv(in => out, σ=identity; aggr=mean, bias=true) — GraphNeuralNetworks/src/layers/conv.jl:770-787"""

    def __init__(self, ch, sigma=None, aggr="mean", bias=True, device="cuda", seed=None):
        cin, cout = ch
        self.weight = glorot_uniform(cout, 2 * cin, device=device, seed=seed)
        self.bias = torch.zeros(cout, dtype=torch.float32, device=device) if bias else None
        self.sigma = sigma
        self.aggr = aggr

    def __call__(self, g, x):
        return sage_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# GATConv
# ---------------------------------------------------------------------------------------------------------
def dropout_keep(seed, p, n_edges, heads, device="cuda"):
    """keep[e][h] in {0, 1} of the attention dropout (include/gnnmp.h: gnnmp_gat_conv_drop_f32): the mask the kernels recompute from
    (seed, e, h), e = 0-based edge position with plan-added self loops at E + node"""
    keep = torch.empty((n_edges, heads), dtype=torch.uint8, device=device)
    L.check(L.load().gnnmp_dropout_keep_u8(int(seed), float(p), n_edges, heads, L.ptr(keep), L.stream_ptr()))
    return keep


def gat_conv(l, g: GNNGraph, x, e=None, return_alpha=False, exact_order=False, seed=None):
    """GNNlib/src/layers/conv.jl:112-167 (no edge features: dense_e === nothing).  dense_x GEMM -> node scores ->
    one fused edge-softmax + weighted aggregate over the (self-looped) plan.  l.dropout > 0: `α = dropout(α, l.dropout)` (conv.jl:139)
    inside the same kernel; `seed` (default: the layer's next seed, see GATConv.next_seed) defines the mask."""
    check_num_nodes(g, x)
    dense_e = getattr(l, "dense_e_weight", None)
    assert not (e is None and dense_e is not None), "Input edge features required for this layer"
    assert not (e is not None and dense_e is None), "Input edge features were not specified in the layer constructor"
    loops = bool(l.add_self_loops)
    if loops:
        assert e is None, "Using edge features and setting add_self_loops=true at the same time is not yet supported."
    plan = g.plan(loops)
    H = l.heads
    C = l.channel[1]
    N = g.num_nodes
    # opt-in (gnnmp/placement.py): Wx = dense_x(x) in a placement class other than x's, the attention output in a class other than both
    # (the best of the candidates by trial) — persistent buffers of the layer; otherwise both are fresh allocations
    out = wx = ch = tok = None
    if (placement.enabled(l) and e is None and not (return_alpha or exact_order) and float(getattr(l, "dropout", 0.0)) == 0.0
            and placement.worth_it((N, H * C))):
        ar = placement.arena()
        if ar is not None:
            cx = ar.class_of(x)
            wx, cw = placement.buffer_for(l, "Wx", (N, H * C), [cx])
            if wx is not None:
                # not Wx's class (the gathered matrix), and with three classes not x's either: in a stack of layers the next kernel usually
                # gathers from x again while this output's dirty lines are still being written back
                ch = placement.choice_for(l, "out", (N, H * C), [cw], prefer_not=[cx])
    Wx = dense(x, l.dense_x_weight, out=wx)                # reshape(dense_x(x), C, H, N)
    a_hc = l.a_hc                                          # [H][2C] (node part)
    lib = L.load()
    if ch is not None:
        out, tok = ch.begin()
    if out is None:
        out = torch.empty((N, H * C), dtype=torch.float32, device=x.device)
    code, post = _act_code(l.sigma)
    fuse_tail = bool(l.concat)
    b = l.bias if (fuse_tail and l.bias is not None) else None
    alpha = None
    p_drop = float(getattr(l, "dropout", 0.0))
    if p_drop > 0.0:
        # like the reference, the layer function drops whenever l.dropout > 0 (NNlib.dropout has no test mode of its own)
        assert e is None and not (return_alpha or exact_order), "attention dropout: one-pass kernel only (no edge features / alpha output)"
        _check_drop_width(H, C, "GATConv")
        if seed is None:
            seed = l.next_seed()
        l.last_seed = int(seed)
        L.check(lib.gnnmp_gat_conv_drop_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), float(l.negative_slope), p_drop, int(seed),
                                            L.ptr(b), code if fuse_tail else L.ACT_IDENTITY, L.ptr(out), None, H, C, L.stream_ptr()))
    elif e is not None:
        # edge features (conv.jl:152-167): We = dense_e(e); the edge's share of the logit a[2C:3C, h] . We_k[:, h] is one
        # scalar per edge and head, added inside the one-pass kernel
        assert not (return_alpha or exact_order), "alpha output / reference order: not with edge features"
        from .graph import check_num_edges
        check_num_edges(g, e)
        We = dense(e, dense_e)
        es = torch.empty((g.num_edges, H), dtype=torch.float32, device=x.device)
        L.check(lib.gnnmp_gat_node_scores_f32(L.ptr(We), L.ptr(l.a_edge_hc), L.ptr(es), None, g.num_edges, H, C, L.stream_ptr()))
        L.check(lib.gnnmp_gat_conv_edge_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), L.ptr(es), float(l.negative_slope),
                                            L.ptr(b), code if fuse_tail else L.ACT_IDENTITY, L.ptr(out), H, C, L.stream_ptr()))
    elif return_alpha or exact_order:
        # the reference's operation order (max pass, denominator pass, α = num/den, β = α .* Wxj) and the α output
        sd = torch.empty((N, H), dtype=torch.float32, device=x.device)
        ss = torch.empty((N, H), dtype=torch.float32, device=x.device)
        L.check(lib.gnnmp_gat_node_scores_f32(L.ptr(Wx), L.ptr(a_hc), L.ptr(sd), L.ptr(ss), N, H, C, L.stream_ptr()))
        alpha = torch.empty((plan.n_total, H), dtype=torch.float32, device=x.device) if return_alpha else None
        L.check(lib.gnnmp_gat_aggregate_f32(plan.handle, L.ptr(Wx), L.ptr(sd), L.ptr(ss), float(l.negative_slope),
                                            L.ptr(b), code if fuse_tail else L.ACT_IDENTITY, L.ptr(out), L.ptr(alpha),
                                            H, C, L.stream_ptr()))
    else:
        # one pass over the edges: in-register logits + online softmax (csrc/gat_fused.hip)
        pr = L._probe
        e0 = pr.begin() if pr is not None else None
        L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), float(l.negative_slope), L.ptr(b),
                                       code if fuse_tail else L.ACT_IDENTITY, L.ptr(out), H, C, L.stream_ptr()))
        if pr is not None:
            pr.end("gat_conv", e0)
        if ch is not None:
            ch.end(tok)
    if fuse_tail:
        y = post(out) if post is not None else out
    else:
        # mean(x, dims = 2) over heads, then σ.(x .+ bias) (conv.jl:143-147): one small kernel
        y = torch.empty((N, C), dtype=torch.float32, device=x.device)
        bb = None if l.bias is None or l.bias is False else l.bias.contiguous()
        L.check(lib.gnnmp_head_mean_f32(L.ptr(out), L.ptr(bb), code, L.ptr(y), N, H, C, L.stream_ptr()))
        y = post(y) if post is not None else y
    return (y, alpha) if return_alpha else y


def _check_drop_width(H, C, name):
    """attention dropout lives inside the one-pass kernel, which needs a feature row that fits one wave (gnnmp.h:
    gnnmp_gat_conv_drop_f32).  The Julia extension falls back to the reference's generic path on wider rows; this host mirror has no
    generic path, so the combination is refused here with a message instead of a GNNMP_EUNSUPPORTED from the C call."""
    vec = 4 if C % 4 == 0 else (2 if C % 2 == 0 else 1)
    if (H * C) // vec > 64:
        raise ValueError(f"{name}(dropout > 0) with heads * out = {H * C} channels: the feature row does not fit one wave "
                         f"({(H * C) // vec} lanes of {vec} floats > 64); attention dropout is only available on the one-pass kernel")


class GATConv:
    """GATConv(in => out, σ=identity; heads=1, concat=true, negative_slope=0.2, bias=true, add_self_loops=true,
    dropout=0.0) — GraphNeuralNetworks/src/layers/conv.jl:309-346.  `a` has the Julia shape (2*out, heads)."""

    def __init__(self, ch, sigma=None, heads=1, concat=True, negative_slope=0.2, bias=True, add_self_loops=True,
                 dropout=0.0, device="cuda", seed=None):
        cin, cout = ch
        ein = 0
        if isinstance(cin, tuple):                         # GATConv((in, ein) => out, ...): edge features of size ein
            cin, ein = cin
        assert 0.0 <= dropout < 1.0
        assert dropout == 0.0 or ein == 0, "attention dropout with edge features is not covered"
        if add_self_loops:
            assert ein == 0, "Using edge features and setting add_self_loops=true at the same time is not yet supported."
        self.dropout = float(dropout)
        self.last_seed = None
        self._seed_rng = np.random.default_rng(None if seed is None else seed + 7)
        self.channel = (cin, cout)
        self.heads = heads
        self.concat = concat
        self.negative_slope = float(negative_slope)
        self.add_self_loops = add_self_loops
        self.dense_x_weight = glorot_uniform(cout * heads, cin, device=device, seed=seed)
        self.dense_e_weight = glorot_uniform(cout * heads, ein, device=device, seed=None if seed is None else seed + 2) \
            if ein > 0 else None
        self.a = glorot_uniform((3 if ein > 0 else 2) * cout, heads, device=device, seed=None if seed is None else seed + 1)
        nb = cout * heads if concat else cout
        self.bias = torch.zeros(nb, dtype=torch.float32, device=device) if bias else None
        self.sigma = sigma

    @property
    def a_hc(self):
        """[H][2C] row-major image of `a` (a[h][0:C] targets, a[h][C:2C] sources); rebuilt only when `a` changes"""
        key = (self.a.data_ptr(), self.a._version)
        if getattr(self, "_a_hc_key", None) != key:
            C = self.channel[1]
            at = self.a.t()                                             # [H][2C] or [H][3C]
            self._a_hc = at[:, :2 * C].contiguous()
            # the edge part a[2C:3C, :] in the [H][2C] image gnnmp_gat_node_scores_f32 reads (its first half)
            self._a_edge_hc = torch.cat([at[:, 2 * C:], torch.zeros_like(at[:, 2 * C:])], 1).contiguous() \
                if at.shape[1] == 3 * C else None
            self._a_hc_key = key
        return self._a_hc

    @property
    def a_edge_hc(self):
        self.a_hc
        return self._a_edge_hc

    def next_seed(self):
        """a fresh 64-bit mask seed per call (the reference draws a fresh mask from its default RNG on every call)"""
        return int(self._seed_rng.integers(0, 2**63 - 1))

    def __call__(self, g, x, e=None):
        return gat_conv(self, g, x, e)


# ---------------------------------------------------------------------------------------------------------
# GlobalPool, GNNChain, Dense
# ---------------------------------------------------------------------------------------------------------
def global_pool(l, g: GNNGraph, x):
    """GNNlib/src/layers/pool.jl:3-5"""
    return reduce_nodes(l.aggr, g, x)


class GlobalPool:
    """GlobalPool(aggr) — GraphNeuralNetworks/src/layers/pool.jl:35-41"""

    def __init__(self, aggr):
        self.aggr = aggr

    def __call__(self, g, x):
        return global_pool(self, g, x)


class Dense:
    """Flux.Dense(in => out, σ) acting on [N, in] features (used as the classifier head of the example models)."""

    def __init__(self, ch, sigma=None, bias=True, device="cuda", seed=None):
        cin, cout = ch
        self.weight = glorot_uniform(cout, cin, device=device, seed=seed)
        self.bias = torch.zeros(cout, dtype=torch.float32, device=device) if bias else None
        self.sigma = sigma

    def __call__(self, x):
        return dense(x, self.weight, self.bias, self.sigma)


class GNNChain:
    """GNNChain(layers...) — GraphNeuralNetworks/src/layers/basic.jl:106-156: graph layers get (g, x), others x."""

    def __init__(self, *layers):
        self.layers = layers
        self._fused = None

    def __call__(self, g, x):
        y = graphconv_chain(self, g, x)      # the whole chain in one kernel when it is GraphConv* -> GlobalPool -> Dense
        if y is not None:
            return y
        for l in self.layers:
            x = l(x) if isinstance(l, Dense) or not _takes_graph(l) else l(g, x)
        return x


class ChainJobs:
    """gnnmp_chain_jobs_t of a batched graph (csrc/graph_chain2.hip): its member graphs packed into wave jobs of <= 64 rows.
    member_stats = (n_rows, largest member, some member empty) when the caller knows them on the host (gnnmp.dataset): the packing then
    runs ON THE DEVICE (gnnmp_chain_jobs_pack: no copy of the sizes to the host, no synchronisation) and the handle is released
    stream-ordered; otherwise the host packing of gnnmp_chain_jobs_create."""

    def __init__(self, seg_ptr, G, member_stats=None):
        import ctypes
        self._h = ctypes.c_void_p()
        self._packed = member_stats is not None
        self._keep = seg_ptr
        self._last_stream = torch.cuda.current_stream()
        if self._packed:
            n_rows, max_graph, has_empty = member_stats
            L.check(L.load().gnnmp_chain_jobs_pack(ctypes.byref(self._h), L.ptr(seg_ptr), G, int(n_rows), int(max_graph),
                                                   1 if has_empty else 0, L.stream_ptr()))
            self.G, self.N, self.max_graph = G, int(n_rows), int(max_graph)
            self._info = None
        else:
            L.check(L.load().gnnmp_chain_jobs_create(ctypes.byref(self._h), L.ptr(seg_ptr), G, L.stream_ptr()))
            self._info = self._read_info()

    @property
    def handle(self):
        # read right before every call that passes the current stream: a device-packed handle is released behind its LAST use
        # (the stream remembered here), whatever stream is current when the garbage collector runs
        self._last_stream = torch.cuda.current_stream()
        return self._h

    def _read_info(self):
        import ctypes
        info = (ctypes.c_int64 * 5)()
        L.check(L.load().gnnmp_chain_jobs_info(self.handle, info))
        self.G, self.N, self.max_graph = info[1], info[2], info[3]
        return info[0], info[4] / 1000.0

    @property
    def njobs(self):
        if self._info is None:
            self._info = self._read_info()      # (device-packed: synchronises the device)
        return self._info[0]

    @property
    def fill(self):
        if self._info is None:
            self._info = self._read_info()
        return self._info[1]

    def export(self):
        """(tab [njobs][64] int32, hdr [32] int32) as the kernel reads them (tests)"""
        cap = max(int(self.G), 1)
        tab = torch.full((cap, 64), -2, dtype=torch.int32, device="cuda")
        hdr = torch.zeros(32, dtype=torch.int32, device="cuda")
        L.check(L.load().gnnmp_chain_jobs_export(self.handle, L.ptr(tab), cap, L.ptr(hdr), L.stream_ptr()))
        return tab, hdr

    def __del__(self):
        try:
            if self._h:
                if self._packed:
                    import ctypes
                    L.load().gnnmp_chain_jobs_release(self._h, ctypes.c_void_p(self._last_stream.cuda_stream))
                else:
                    L.load().gnnmp_chain_jobs_destroy(self._h)
                self._h = None
        except Exception:
            pass


def chain_prepare(g):
    """the fused chain's wave jobs of a batched graph, made ahead of the model call (DataLoader(prepare = chain_prepare): on the loader's
    side stream, beside the previous step's kernels); the model call finds them in the batch's cache"""
    if g._cache.get("chain_jobs") is None and g._cache.get("node_ptr") is not None:
        g._cache["chain_jobs"] = ChainJobs(g._cache["node_ptr"], g.num_graphs, g._cache.get("member_stats"))
    return g


def _chain_pattern(layers):
    """(convs, pool, head) if `layers` is GraphConv, ..., GraphConv, GlobalPool(+ | mean), Dense(identity) — the graph-classification
    chain of examples/graph_classification_tudataset.jl:79-82 — inside the fused kernel's envelope; else None."""
    if len(layers) < 3 or not isinstance(layers[-1], Dense) or not isinstance(layers[-2], GlobalPool):
        return None
    convs, pool, head = layers[:-2], layers[-2], layers[-1]
    if not (1 <= len(convs) <= 4) or not all(isinstance(c, GraphConv) for c in convs):
        return None
    if pool.aggr not in ("+", "sum", "mean") or head.sigma not in (None, "identity") or head.weight.shape[0] > 8:
        return None
    aggr = convs[0].aggr
    if aggr not in ("+", "sum", "mean") or any(c.aggr != aggr for c in convs):
        return None
    # (the FIRST layer's input width may be anything: graphconv_chain zero-pads the features and that layer's weight columns — TUDataset
    # node features are one-hot labels: MUTAG 7, PROTEINS 3)
    if any(c.sigma not in _ACT or c.weight1.shape[0] > 128 or c.weight1.shape[0] % 4 for c in convs):
        return None
    if any(c.weight1.shape[1] % 4 for c in convs[1:]) or convs[0].weight1.shape[1] > 512:
        return None
    if any(convs[k + 1].weight1.shape[1] != convs[k].weight1.shape[0] for k in range(len(convs) - 1)):
        return None
    if any(c.weight1.shape[0] < 8 for c in convs[:-1]):      # a stored hidden layer of 4 columns is outside the kernel's envelope
        return None
    if head.weight.shape[1] != convs[-1].weight1.shape[0]:
        return None
    return convs, pool, head


def graphconv_chain(model, g: GNNGraph, x):
    """GNNChain(GraphConv..., GlobalPool, Dense) on a batched graph through gnnmp_graphconv_chain_f32 (csrc/graph_chain.hip): one
    launch, layer outputs never leave the workgroup that owns the member graphs.  Returns None when the chain or the graph is not
    of that form (the caller runs it layer by layer)."""
    import ctypes
    if g.graph_indicator is None or x is None or x.dim() != 2 or x.dtype != torch.float32:
        return None
    pat = _chain_pattern(model.layers)
    if pat is None:
        return None
    convs, pool, head = pat
    if x.shape[1] != convs[0].weight1.shape[1] or x.shape[0] < 32:
        return None
    check_num_nodes(g, x)
    lib = L.load()
    nl = len(convs)
    # the ctypes argument block only changes when a weight tensor is replaced: cached on the model
    tensors = [t for c in convs for t in (c.weight1, c.weight2, c.bias)] + [head.weight, head.bias]
    key = tuple((0 if t is None or t is False else (t.data_ptr(), t._version)) for t in tensors) + tuple(c.sigma for c in convs)
    if model._fused is None or model._fused[0] != key:
        keep = [t.contiguous() if isinstance(t, torch.Tensor) else None for t in tensors]
        # Input width of the first layer: padded with zero columns (exact: the extra products are 0 * 0) to 16 when that makes the chain
        # the wave-pair kernel's shape (16 => 128 => 128, csrc/graph_chain2.hip), else to the next multiple of 4 (the general kernel)
        din = convs[0].weight1.shape[1]
        widths = [c.weight1.shape[0] for c in convs]
        din_p = 16 if (nl == 2 and widths == [128, 128] and din <= 16) else (din + 3) // 4 * 4
        if din_p != din:
            for k in (0, 1):
                wp = torch.zeros((widths[0], din_p), dtype=torch.float32, device=keep[k].device)
                wp[:, :din] = keep[k]
                keep[k] = wp
        i64, vp = ctypes.c_int64, ctypes.c_void_p
        dims = (i64 * (nl + 1))(din_p, *widths)
        wr = (vp * nl)(*[keep[3 * k].data_ptr() for k in range(nl)])
        wa = (vp * nl)(*[keep[3 * k + 1].data_ptr() for k in range(nl)])
        bs = (vp * nl)(*[(keep[3 * k + 2].data_ptr() if keep[3 * k + 2] is not None else None) for k in range(nl)])
        act = (ctypes.c_int * nl)(*[_ACT[c.sigma] for c in convs])
        model._fused = (key, keep, dims, wr, wa, bs, act)
    _, keep, dims, wr, wa, bs, act = model._fused
    N, G = g.num_nodes, g.num_graphs
    sp = g._cache.get("node_ptr")
    if sp is None:
        gi = g.graph_indicator
        sp = torch.empty(G + 1, dtype=torch.int64, device=x.device)
        L.check(lib.gnnmp_segment_bounds(L.ptr(gi), 8 if gi.dtype == torch.int64 else 4, g.index_base, N, G, L.ptr(sp), L.stream_ptr()))
        g._cache["node_ptr"] = sp
    nout = head.weight.shape[0]
    jobs = g._cache.get("chain_jobs")
    if jobs is None:
        jobs = ChainJobs(sp, G, g._cache.get("member_stats"))      # member graphs packed into wave jobs: a constant of the batch, like its plan
        g._cache["chain_jobs"] = jobs
    need = lib.gnnmp_graphconv_chain_scratch_floats(N, nl, dims, nout)
    scratch = getattr(model, "_scratch", None)      # (on the model, grow-only: a training loop meets a new batch object every step)
    if scratch is None or scratch.numel() < need or scratch.device != x.device:
        scratch = torch.empty(need, dtype=torch.float32, device=x.device)
        model._scratch = scratch
    out = torch.empty((G, nout), dtype=torch.float32, device=x.device)
    xc = x.contiguous()
    if dims[0] != x.shape[1]:                  # (zero-padded feature columns, see above: one strided copy)
        xc = torch.zeros((N, dims[0]), dtype=torch.float32, device=x.device)
        xc[:, : x.shape[1]] = x
    rc = lib.gnnmp_graphconv_chain_f32(g.plan(False).handle, jobs.handle, L.ptr(sp), G, L.ptr(xc), nl, dims, wr, wa, bs, act, 0,
                                       aggr_code(convs[0].aggr), aggr_code(pool.aggr), L.ptr(keep[-2]), L.ptr(keep[-1]), nout,
                                       L.ptr(scratch), L.ptr(out), L.stream_ptr())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc)
    return out


def _takes_graph(l):
    return isinstance(l, (GCNConv, GraphConv, SAGEConv, GATConv, GlobalPool)) or getattr(l, "takes_graph", False)


def glorot_uniform(rows, cols, device="cuda", seed=None):
    """Flux.glorot_uniform: U(-s, s), s = sqrt(24 / (fan_in + fan_out)) / 2... i.e. sqrt(6 / (rows + cols))"""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(0 if seed is None else int(seed))
    s = math.sqrt(6.0 / (rows + cols))
    w = (torch.rand((rows, cols), generator=gen, dtype=torch.float32) * 2 - 1) * s
    return w.to(device)
