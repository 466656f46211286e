"""The other layers of the reference that run on the same path: GATv2Conv, AGNNConv,
TransformerConv (attention core + root weight + skip connection) and GINConv.  Their attention is the SAME one-pass
kernel as GATConv's with a different per-edge logit (gnnmp_attn_conv_f32, csrc/gat_fused.hip); nothing here does
arithmetic on the host — torch allocates, libgnnmp computes.

Reference bodies: GNNlib/src/layers/conv.jl  gatv2_conv :171-214, gin_conv :250-256, agnn_conv :337-352,
transformer_conv :553-629.  Constructors: GraphNeuralNetworks/src/layers/conv.jl :435-462 (GATv2Conv), :637 (GINConv),
:998-1000 (AGNNConv), :1501-1536 (TransformerConv).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from .graph import GNNGraph, check_num_nodes
from .layers import _act_code, dense, glorot_uniform
from .msgpass import _fused

ATTN_GAT, ATTN_GATV2, ATTN_DOT, ATTN_COS = 0, 1, 2, 3


def attn_conv(plan, mode, K, Q=None, V=None, a=None, slope=0.2, scale=1.0, bias=None, act=L.ACT_IDENTITY, H=1, C=None,
              stats=None, dropout=0.0, seed=0):
    """out[i] = Σ_j softmax_{j in N(i)}(logit_mode(Q_i, K_j)) V_j  (+ bias, act) — one pass over the plan's edges; dropout > 0: the
    coefficients dropped inside the kernel with the mask of `seed` (include/gnnmp.h: gnnmp_gat_conv_drop_f32)"""
    N = plan.n_dst
    C = K.shape[1] // H if C is None else C
    out = torch.empty((N, H * C), dtype=torch.float32, device=K.device)
    if dropout > 0.0:
        L.check(L.load().gnnmp_attn_conv_drop_f32(plan.handle, mode, L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(a), float(slope),
                                                  float(scale), float(dropout), int(seed), L.ptr(bias), act, L.ptr(out),
                                                  L.ptr(stats), H, C, L.stream_ptr()))
    else:
        L.check(L.load().gnnmp_attn_conv_f32(plan.handle, mode, L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(a), float(slope),
                                             float(scale), L.ptr(bias), act, L.ptr(out), L.ptr(stats), H, C, L.stream_ptr()))
    return out


def _heads_tail(out, l, N, H, C):
    """concat = true: bias and σ were fused into the kernel; concat = false: mean over heads, then σ.(x .+ bias)"""
    code, post = _act_code(l.sigma)
    if l.concat:
        return post(out) if post is not None else out
    y = torch.empty((N, C), dtype=torch.float32, device=out.device)
    bb = None if l.bias is None or l.bias is False else l.bias.contiguous()
    L.check(L.load().gnnmp_head_mean_f32(L.ptr(out), L.ptr(bb), code, L.ptr(y), N, H, C, L.stream_ptr()))
    return post(y) if post is not None else y


# ---------------------------------------------------------------------------------------------------------
# GATv2Conv
# ---------------------------------------------------------------------------------------------------------
def gatv2_conv(l, g: GNNGraph, x, e=None, seed=None):
    """conv.jl:171-214 without edge features: logα = sum(a .* leakyrelu.(Wxi + Wxj)), softmax over the neighbourhood,
    `α = dropout(α, l.dropout)` (:191, inside the kernel; `seed` defaults to the layer's next one), weighted sum of Wxj, optional
    head mean, σ.(x .+ bias)"""
    check_num_nodes(g, x)
    assert e is None and l.dense_e is None, "edge features (dense_e) are outside the hot path"
    plan = g.plan(bool(l.add_self_loops))
    H, C = l.heads, l.channel[1]
    Wxi = dense(x, l.dense_i_weight, l.dense_i_bias)
    Wxj = dense(x, l.dense_j_weight)
    code, _ = _act_code(l.sigma)
    fuse = bool(l.concat)
    b = l.bias if (fuse and l.bias is not None) else None
    p_drop = float(getattr(l, "dropout", 0.0))
    if p_drop > 0.0:
        from .layers import _check_drop_width
        _check_drop_width(H, C, "GATv2Conv")
        if seed is None:
            seed = l.next_seed()
        l.last_seed = int(seed)
    out = attn_conv(plan, ATTN_GATV2, Wxj, Q=Wxi, a=l.a_hc, slope=l.negative_slope, bias=b,
                    act=code if fuse else L.ACT_IDENTITY, H=H, C=C, dropout=p_drop, seed=0 if seed is None else seed)
    return _heads_tail(out, l, g.num_nodes, H, C)


class GATv2Conv:
    """GATv2Conv(in => out, σ=identity; heads=1, concat=true, negative_slope=0.2, bias=true, add_self_loops=true) —
    GraphNeuralNetworks/src/layers/conv.jl:435-462.  `a` has the Julia shape (out, heads); dense_i carries a bias iff
    `bias`, dense_j none."""

    takes_graph = True

    def __init__(self, ch, sigma=None, heads=1, concat=True, negative_slope=0.2, bias=True, add_self_loops=True,
                 dropout=0.0, device="cuda", seed=None):
        cin, cout = ch
        assert 0.0 <= dropout < 1.0
        sd = (lambda k: None if seed is None else seed + k)
        self.dropout = float(dropout)
        self.last_seed = None
        self._seed_rng = np.random.default_rng(None if seed is None else seed + 7)
        self.channel = (cin, cout)
        self.heads, self.concat, self.negative_slope = heads, concat, float(negative_slope)
        self.add_self_loops = add_self_loops
        self.dense_i_weight = glorot_uniform(cout * heads, cin, device=device, seed=sd(0))
        self.dense_i_bias = torch.zeros(cout * heads, dtype=torch.float32, device=device) if bias else None
        self.dense_j_weight = glorot_uniform(cout * heads, cin, device=device, seed=sd(1))
        self.dense_e = None
        self.a = glorot_uniform(cout, heads, device=device, seed=sd(2))
        self.bias = torch.zeros(cout * heads if concat else cout, dtype=torch.float32, device=device) if bias else None
        self.sigma = sigma

    @property
    def a_hc(self):
        key = (self.a.data_ptr(), self.a._version)
        if getattr(self, "_a_hc_key", None) != key:
            self._a_hc = self.a.t().contiguous()          # [H][C]
            self._a_hc_key = key
        return self._a_hc

    def next_seed(self):
        """a fresh 64-bit mask seed per call (the reference draws a fresh mask from its default RNG on every call)"""
        return int(self._seed_rng.integers(0, 2**63 - 1))

    def __call__(self, g, x, e=None):
        return gatv2_conv(self, g, x, e)


# ---------------------------------------------------------------------------------------------------------
# AGNNConv
# ---------------------------------------------------------------------------------------------------------
def agnn_conv(l, g: GNNGraph, x):
    """conv.jl:337-352: α = softmax_edge_neighbors(β .* cos(x_i, x_j)); out_i = Σ_j α_ij x_j.  The norms are formed in
    registers from the rows the aggregation fetches anyway (no normalised copy of x, no (1, E') cosine array)."""
    check_num_nodes(g, x)
    plan = g.plan(bool(l.add_self_loops))
    x = x.contiguous()
    return attn_conv(plan, ATTN_COS, x, scale=l.beta, H=1, C=x.shape[1])


class AGNNConv:
    """AGNNConv(; init_beta=1, add_self_loops=true, trainable=true) — GraphNeuralNetworks/src/layers/conv.jl:998-1000"""

    takes_graph = True

    def __init__(self, init_beta=1.0, add_self_loops=True, trainable=True):
        self.beta = float(init_beta)
        self.add_self_loops = add_self_loops
        self.trainable = trainable

    def __call__(self, g, x):
        return agnn_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# TransformerConv
# ---------------------------------------------------------------------------------------------------------
def transformer_conv(l, g: GNNGraph, x, e=None):
    """conv.jl:553-629 for the configuration without edge features, gating, batch norm and feed-forward block:
    α = softmax((W3 x_i) . (W4 x_j) / sqrt(out)), h_i = Σ_j α_ij W2 x_j, [mean over heads], + W1 x_i, + x_i"""
    check_num_nodes(g, x)
    assert e is None, "edge features (W6) are outside the hot path"
    plan = g.plan(bool(l.add_self_loops))
    H, C = l.heads, l.channels[1]
    x = x.contiguous()
    W2x = dense(x, l.W2_weight, l.W2_bias)
    W3x = dense(x, l.W3_weight, l.W3_bias)
    W4x = dense(x, l.W4_weight, l.W4_bias)
    h = attn_conv(plan, ATTN_DOT, W4x, Q=W3x, V=W2x, scale=l.sqrt_out, H=H, C=C)
    lib = L.load()
    N = g.num_nodes
    if not l.concat:
        y = torch.empty((N, C), dtype=torch.float32, device=x.device)
        L.check(lib.gnnmp_head_mean_f32(L.ptr(h), None, L.ACT_IDENTITY, L.ptr(y), N, H, C, L.stream_ptr()))
        h = y
    if l.W1_weight is not None:
        W1x = dense(x, l.W1_weight, l.W1_bias)
        L.check(lib.gnnmp_add_f32(L.ptr(h), L.ptr(W1x), L.ptr(h), h.numel(), L.stream_ptr()))
    if l.skip_connection:
        assert h.shape[1] == x.shape[1], "In-channels must correspond to out-channels * heads if skip_connection is used"
        L.check(lib.gnnmp_add_f32(L.ptr(h), L.ptr(x), L.ptr(h), h.numel(), L.stream_ptr()))
    return h


class TransformerConv:
    """TransformerConv(in => out; heads=1, concat=true, add_self_loops=false, bias_qkv=true, bias_root=true,
    root_weight=true, skip_connection=false) — GraphNeuralNetworks/src/layers/conv.jl:1501-1536 (gating, batch_norm,
    ff_channels and edge features are not on the hot path and are rejected)"""

    takes_graph = True

    def __init__(self, ch, heads=1, concat=True, add_self_loops=False, bias_qkv=True, bias_root=True, root_weight=True,
                 gating=False, skip_connection=False, batch_norm=False, ff_channels=0, device="cuda", seed=None):
        cin, cout = ch
        assert not gating and not batch_norm and ff_channels == 0, "gating / batch_norm / ff block: outside the hot path"
        sd = (lambda k: None if seed is None else seed + k)
        z = (lambda nrow, on: torch.zeros(nrow, dtype=torch.float32, device=device) if on else None)
        self.channels = (cin, cout)
        self.heads, self.concat, self.add_self_loops, self.skip_connection = heads, concat, add_self_loops, skip_connection
        out_mha = cout * (heads if concat else 1)
        self.W1_weight = glorot_uniform(out_mha, cin, device=device, seed=sd(0)) if root_weight else None
        self.W1_bias = z(out_mha, bias_root and root_weight)
        self.W2_weight = glorot_uniform(cout * heads, cin, device=device, seed=sd(1))
        self.W3_weight = glorot_uniform(cout * heads, cin, device=device, seed=sd(2))
        self.W4_weight = glorot_uniform(cout * heads, cin, device=device, seed=sd(3))
        self.W2_bias, self.W3_bias, self.W4_bias = (z(cout * heads, bias_qkv) for _ in range(3))
        self.sqrt_out = float(torch.tensor(float(cout), dtype=torch.float32).sqrt())   # Float32(√out)

    def __call__(self, g, x, e=None):
        return transformer_conv(self, g, x, e)


# ---------------------------------------------------------------------------------------------------------
# GINConv
# ---------------------------------------------------------------------------------------------------------
def gin_conv(l, g: GNNGraph, x):
    """conv.jl:250-256: nn((1 + ϵ) .* xi .+ propagate(copy_xj, g, aggr; xj))"""
    check_num_nodes(g, x)
    x = x.contiguous()
    m = _fused(g, L.COPY_XJ, l.aggr, x, None)
    z = torch.empty_like(m)
    L.check(L.load().gnnmp_axpy_f32(float(torch.tensor(1.0, dtype=torch.float32) + torch.tensor(l.eps, dtype=torch.float32)),
                                    L.ptr(x), L.ptr(m), L.ptr(z), z.numel(), L.stream_ptr()))
    return l.nn(z)


class GINConv:
    """GINConv(nn, ϵ; aggr=+) — GraphNeuralNetworks/src/layers/conv.jl:637; `nn` is any callable on [N, D] (gnnmp.Dense,
    a GNNChain of them, ...)"""

    takes_graph = True

    def __init__(self, nn, eps, aggr="+"):
        self.nn, self.eps, self.aggr = nn, float(eps), aggr

    def __call__(self, g, x):
        return gin_conv(self, g, x)


__all__ = ["GATv2Conv", "AGNNConv", "TransformerConv", "GINConv", "gatv2_conv", "agnn_conv", "transformer_conv",
           "gin_conv", "attn_conv", "ATTN_GAT", "ATTN_GATV2", "ATTN_DOT", "ATTN_COS"]
