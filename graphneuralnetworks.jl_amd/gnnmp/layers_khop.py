"""k-hop layers that are repeated applications of the GCN-normalised propagate: SGConv and TAGConv.
GNNlib/src/layers/conv.jl  sg_conv :501-542, tag_conv :634-685; constructors GraphNeuralNetworks/src/layers/conv.jl
(SGConv(in => out, k = 1; bias, add_self_loops = true, use_edge_weight = false), TAGConv(in => out, k = 3; ...)).
Every hop is ONE fused kernel launch (both `x .* c'` scalings folded into the aggregation, per-edge coefficients cached in
plan order on the graph); nothing is computed on the host.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .graph import GNNGraph, check_num_nodes
from .layers import dense, glorot_uniform


def _norm_slots(g: GNNGraph, loops: bool, w):
    """(c, ss_slot, w_slot): c = 1 ./ sqrt.(degree(g; dir = :in, edge_weight)), the source factor and the edge weight in plan
    slot order.  Constants of the graph when the weights are the graph's own (or absent): cached on it like gcn_conv's.  An
    `edge_weight` ARGUMENT is never cached — its address and version say nothing about its contents once the caller frees it
    and the allocator hands the same block to the next weight vector (and every distinct key would pin 2 E' floats)."""
    from .layers import _inv_sqrt
    cacheable = w is None or w is g.w
    key = ("gcn_norm_slots", bool(loops), w is not None)
    hit = g._cache.get(key) if cacheable else None
    if hit is not None:
        return hit
    lib = L.load()
    plan = g.plan(loops)
    N = g.num_nodes
    d = torch.empty(N, dtype=torch.float32, device=g.device)
    L.check(lib.gnnmp_degree_f32(plan.handle, L.ptr(w), L.ptr(d), L.stream_ptr()))
    c = _inv_sqrt(d)
    ss = torch.empty(plan.n_total, dtype=torch.float32, device=g.device)
    L.check(lib.gnnmp_plan_slot_gather_f32(plan.handle, 0, L.ptr(c), L.ptr(ss), L.stream_ptr()))
    ws = None
    if w is not None:
        ws = torch.empty(plan.n_total, dtype=torch.float32, device=g.device)
        L.check(lib.gnnmp_plan_slot_gather_f32(plan.handle, 1, L.ptr(w), L.ptr(ws), L.stream_ptr()))
    if cacheable:
        g._cache[key] = (c, ss, ws)
    return c, ss, ws


def _hop(g: GNNGraph, loops: bool, x, c, ss, ws):
    """x .* c' -> propagate(copy_xj | w_mul_xj, g, +) -> .* c'   (conv.jl:528-536)"""
    plan = g.plan(loops)
    out = torch.empty((plan.n_dst, x.shape[1]), dtype=torch.float32, device=x.device)
    L.check(L.load().gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(x), L.ptr(ws), L.ptr(ss), L.ptr(c), L.ptr(out),
                                               x.shape[1], L.stream_ptr()))
    return out


def _edge_weights(l, g: GNNGraph, edge_weight):
    if edge_weight is not None:
        assert edge_weight.numel() == g.num_edges, \
            f"Wrong number of edge weights (expected {g.num_edges} but given {edge_weight.numel()})"
        return edge_weight.to(torch.float32).contiguous()
    return g.w if (l.use_edge_weight and g.w is not None) else None


def sg_conv(l, g: GNNGraph, x, edge_weight=None):
    """conv.jl:501-542: W applied first if it shrinks the features, k normalised hops, W last otherwise, + bias"""
    check_num_nodes(g, x)
    w = _edge_weights(l, g, edge_weight)
    loops = bool(l.add_self_loops)
    c, ss, ws = _norm_slots(g, loops, w)
    Dout, Din = l.weight.shape
    x = x.contiguous()
    if Dout < Din:
        x = dense(x, l.weight)
    for _ in range(l.k):
        x = _hop(g, loops, x, c, ss, ws)
    if Dout >= Din:
        return dense(x, l.weight, l.bias)
    from .layers import bias_act
    return bias_act(x, l.bias, None)


def tag_conv(l, g: GNNGraph, x, edge_weight=None):
    """conv.jl:634-685: sum_total = Σ_{iter} W * (Σ_{j <= iter} Ã^j x), accumulated in the reference's order, + bias"""
    check_num_nodes(g, x)
    w = _edge_weights(l, g, edge_weight)
    loops = bool(l.add_self_loops)
    c, ss, ws = _norm_slots(g, loops, w)
    lib = L.load()
    x = x.contiguous()
    sum_pow = sum_total = None
    for it in range(l.k):
        x = _hop(g, loops, x, c, ss, ws)
        if it == 0:
            sum_pow = x
            sum_total = dense(sum_pow, l.weight)
        else:
            nxt = torch.empty_like(sum_pow)
            L.check(lib.gnnmp_add_f32(L.ptr(sum_pow), L.ptr(x), L.ptr(nxt), nxt.numel(), L.stream_ptr()))
            sum_pow = nxt
            term = dense(sum_pow, l.weight)
            L.check(lib.gnnmp_add_f32(L.ptr(sum_total), L.ptr(term), L.ptr(sum_total), term.numel(), L.stream_ptr()))
    from .layers import bias_act
    return bias_act(sum_total, l.bias, None)


class _KHop:
    takes_graph = True

    def __init__(self, ch, k, bias=True, add_self_loops=True, use_edge_weight=False, device="cuda", seed=None):
        cin, cout = ch
        self.weight = glorot_uniform(cout, cin, device=device, seed=seed)
        self.bias = torch.zeros(cout, dtype=torch.float32, device=device) if bias else None
        self.k, self.add_self_loops, self.use_edge_weight = int(k), add_self_loops, use_edge_weight


class SGConv(_KHop):
    """SGConv(in => out, k = 1; bias = true, add_self_loops = true, use_edge_weight = false)"""

    def __init__(self, ch, k=1, **kw):
        super().__init__(ch, k, **kw)

    def __call__(self, g, x, edge_weight=None):
        return sg_conv(self, g, x, edge_weight)


class TAGConv(_KHop):
    """TAGConv(in => out, k = 3; bias = true, add_self_loops = true, use_edge_weight = false)"""

    def __init__(self, ch, k=3, **kw):
        super().__init__(ch, k, **kw)

    def __call__(self, g, x, edge_weight=None):
        return tag_conv(self, g, x, edge_weight)


# ---------------------------------------------------------------------------------------------------------
# ResGatedGraphConv: the gated message  sigmoid.(A*xi .+ B*xj) .* (V*xj)  as a functor of the fused propagate
# ---------------------------------------------------------------------------------------------------------
def res_gated_graph_conv(l, g: GNNGraph, x):
    """GNNlib/src/layers/conv.jl:287-300: σ.(U*xi .+ Σ_j sigmoid.(A*xi .+ B*xj) .* V*xj .+ bias).  B and V are applied by
    ONE dense call on the stacked weight, so that an edge reads its source's (Bx, Vx) as one contiguous row."""
    from .layers import bias_act
    check_num_nodes(g, x)
    x = x.contiguous()
    lib = L.load()
    plan = g.plan(False)
    D = l.A.shape[0]
    Ax = dense(x, l.A)
    BVx = dense(x, l.BV)                                              # [N][2D] = [Bx | Vx]
    m = torch.empty((plan.n_dst, D), dtype=torch.float32, device=x.device)
    L.check(lib.gnnmp_propagate_gated_f32(plan.handle, L.SUM, L.ptr(Ax), L.ptr(BVx), L.ptr(m), D, L.stream_ptr()))
    Ux = dense(x, l.U)
    L.check(lib.gnnmp_add_f32(L.ptr(Ux), L.ptr(m), L.ptr(Ux), Ux.numel(), L.stream_ptr()))     # (U*xi .+ m)
    return bias_act(Ux, l.bias, l.sigma)


class ResGatedGraphConv:
    """ResGatedGraphConv(in => out, σ = identity; bias = true) — GraphNeuralNetworks/src/layers/conv.jl:849-858"""

    takes_graph = True

    def __init__(self, ch, sigma=None, bias=True, device="cuda", seed=None):
        cin, cout = ch
        sd = (lambda k: None if seed is None else seed + k)
        self.A = glorot_uniform(cout, cin, device=device, seed=sd(0))
        self.B = glorot_uniform(cout, cin, device=device, seed=sd(1))
        self.U = glorot_uniform(cout, cin, device=device, seed=sd(2))
        self.V = glorot_uniform(cout, cin, device=device, seed=sd(3))
        self.bias = torch.zeros(cout, dtype=torch.float32, device=device) if bias else None
        self.sigma = sigma

    @property
    def BV(self):
        """[B; V] stacked (2*out, in); rebuilt only when B or V change"""
        key = (self.B.data_ptr(), self.B._version, self.V.data_ptr(), self.V._version)
        if getattr(self, "_bv_key", None) != key:
            self._bv = torch.cat([self.B, self.V], 0).contiguous()
            self._bv_key = key
        return self._bv

    def __call__(self, g, x):
        return res_gated_graph_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# GlobalAttentionPool: softmax_nodes(g, fgate(x)) .* ffeat(x), summed per graph
# ---------------------------------------------------------------------------------------------------------
def global_attention_pool(l, g: GNNGraph, x):
    """GNNlib/src/layers/pool.jl:6-10.  fgate maps [N, D] -> [N, 1] (or [N, D']); ffeat -> [N, D'] (identity if None)."""
    from .utils import reduce_nodes, softmax_nodes
    check_num_nodes(g, x)
    alpha = softmax_nodes(g, l.fgate(x)).contiguous()
    feats = (x if l.ffeat is None else l.ffeat(x)).contiguous()
    assert alpha.shape[1] in (1, feats.shape[1]), "the gate must have one channel or as many as the features"
    out = torch.empty_like(feats)
    L.check(L.load().gnnmp_mul_rows_f32(L.ptr(alpha), alpha.shape[1], L.ptr(feats), L.ptr(out), feats.shape[0], feats.shape[1],
                                        L.stream_ptr()))
    return reduce_nodes("+", g, out)


class GlobalAttentionPool:
    """GlobalAttentionPool(fgate, ffeat = identity) — GraphNeuralNetworks/src/layers/pool.jl"""

    takes_graph = True

    def __init__(self, fgate, ffeat=None):
        self.fgate, self.ffeat = fgate, ffeat

    def __call__(self, g, x):
        return global_attention_pool(self, g, x)
