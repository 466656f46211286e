"""CPU oracle for the GNNlib.jl message-passing hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module, and only
as the checker / the reported CPU baseline.  The product (``graphneuralnetworks.jl_amd/``) never does.

It wraps ``libgnn_oracle.so`` (``gnn_oracle.c``: the loops whose ORDER matters — gather, scatter, CSC SpMM, GAT
logits) and composes the four layer bodies in numpy exactly in the order the reference does.  Pin status and the
list of reference known-answer tests that pin it: see the header of ``gnn_oracle.c``.

Conventions: feature arrays are ``float32`` C-contiguous ``[N, D]`` (= Julia ``(D, N)`` column-major); index
vectors are ``int64`` and 1-based, as Julia holds them.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgnn_oracle.so")

SUM, MEAN, MAX, MIN = 0, 1, 2, 3
_AGGR = {"+": SUM, "sum": SUM, "add": SUM, "mean": MEAN, "max": MAX, "min": MIN,
         SUM: SUM, MEAN: MEAN, MAX: MAX, MIN: MIN}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gnn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_gather.restype = ctypes.c_int
        _lib.orc_scatter.restype = ctypes.c_int
        _lib.orc_degree.restype = ctypes.c_int
        _lib.orc_propagate.restype = ctypes.c_int
        for name in ("orc_gather64", "orc_scatter64", "orc_propagate64"):
            getattr(_lib, name).restype = ctypes.c_int
        _lib.orc_spmm_csc.restype = ctypes.c_int
        _lib.orc_softmax_edge_neighbors.restype = ctypes.c_int
        for name in ("orc_add_self_loops", "orc_batch", "orc_gat_logits", "orc_gat_weight_messages",
                     "orc_scale_rows", "orc_inv_sqrt", "orc_matmul"):
            getattr(_lib, name).restype = None
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i(v):
    return ctypes.c_int64(int(v))


def _check(rc, what):
    if rc == -1:
        raise IndexError(f"{what}: index out of range")
    if rc != 0:
        raise MemoryError(f"{what}: rc={rc}")


# ---------------------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------------------
def _is64(a):
    return getattr(a, "dtype", None) == np.float64


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def gather(x, idx):
    """NNlib.gather — GNNGraphs/src/gatherscatter.jl:4.  x [N, ...] -> [K, ...]  (Float64 input: the Float64 loop)"""
    if _is64(x):
        x = _f64(x)
        idx = _i64(idx)
        D = int(np.prod(x.shape[1:], dtype=np.int64))
        out = np.empty((idx.shape[0],) + x.shape[1:], np.float64)
        _check(lib().orc_gather64(_p(x), _i(x.shape[0]), _i(D), _p(idx), _i(idx.shape[0]), _p(out)), "gather")
        return out
    x = _f32(x)
    idx = _i64(idx)
    n = x.shape[0]
    D = int(np.prod(x.shape[1:], dtype=np.int64))
    out = np.empty((idx.shape[0],) + x.shape[1:], np.float32)
    _check(lib().orc_gather(_p(x), _i(n), _i(D), _p(idx), _i(idx.shape[0]), _p(out)), "gather")
    return out


def scatter(aggr, src, idx, n=None):
    """NNlib.scatter(aggr, src, idx; dstsize) — GNNGraphs/src/gatherscatter.jl:12-18.
    n omitted => maximum(idx), like NNlib (GNNlib/src/utils.jl:15).  Float64 input: the Float64 loop."""
    if _is64(src):
        src = _f64(src)
        idx = _i64(idx)
        assert src.shape[0] == idx.shape[0]
        if n is None:
            n = int(idx.max()) if idx.size else 0
        D = int(np.prod(src.shape[1:], dtype=np.int64))
        out = np.empty((n,) + src.shape[1:], np.float64)
        _check(lib().orc_scatter64(_AGGR[aggr], _p(src), _i(D), _p(idx), _i(idx.shape[0]), _i(n), _p(out)), "scatter")
        return out
    src = _f32(src)
    idx = _i64(idx)
    assert src.shape[0] == idx.shape[0]
    if n is None:
        n = int(idx.max()) if idx.size else 0
    D = int(np.prod(src.shape[1:], dtype=np.int64))
    out = np.empty((n,) + src.shape[1:], np.float32)
    _check(lib().orc_scatter(_AGGR[aggr], _p(src), _i(D), _p(idx), _i(idx.shape[0]), _i(n), _p(out)), "scatter")
    return out


def degree(idx, n, w=None):
    """degree(g, Float32; dir, edge_weight) — GNNGraphs/src/query.jl:355-369 (idx = t for :in, s for :out)."""
    idx = _i64(idx)
    w = None if w is None else _f32(w)
    out = np.empty((n,), np.float32)
    _check(lib().orc_degree(_p(idx), _p(w), _i(idx.shape[0]), _i(n), _p(out)), "degree")
    return out


def add_self_loops(s, t, n, w=None):
    """add_self_loops(g::GNNGraph{COO}) — GNNGraphs/src/transform.jl:12-28."""
    s = _i64(s)
    t = _i64(t)
    E = s.shape[0]
    s2 = np.empty(E + n, np.int64)
    t2 = np.empty(E + n, np.int64)
    w = None if w is None else _f32(w)
    w2 = None if w is None else np.empty(E + n, np.float32)
    lib().orc_add_self_loops(_p(s), _p(t), _p(w), _i(E), _i(n), _p(s2), _p(t2), _p(w2))
    return s2, t2, w2


def batch(graphs):
    """MLUtils.batch(::Vector{GNNGraph{COO}}) — GNNGraphs/src/transform.jl:682-709.
    graphs: list of (s, t, num_nodes).  Returns s, t, graph_indicator, num_nodes."""
    ne = np.array([0] + [len(g[0]) for g in graphs], np.int64).cumsum()
    nn = np.array([0] + [int(g[2]) for g in graphs], np.int64).cumsum()
    s = _i64(np.concatenate([_i64(g[0]) for g in graphs])) if graphs else np.zeros(0, np.int64)
    t = _i64(np.concatenate([_i64(g[1]) for g in graphs])) if graphs else np.zeros(0, np.int64)
    s2 = np.empty_like(s)
    t2 = np.empty_like(t)
    gi = np.empty(int(nn[-1]), np.int64)
    lib().orc_batch(_p(s), _p(t), _p(ne), _p(nn), _i(len(graphs)), _p(s2), _p(t2), _p(gi))
    return s2, t2, gi, int(nn[-1])


def propagate(aggr, s, t, n, xj, w=None, n_dst=None):
    """Generic propagate (gather -> message -> scatter) with copy_xj (w None) or w_mul_xj / e_mul_xj (vector w)
    — GNNlib/src/msgpass.jl:71-79,121-129,145-149,162,191-208.  This is what the reference runs for any aggr and,
    on GPU arrays, also for `+` (GNNlib/ext/GNNlibAMDGPUExt.jl:13-32).  Float64 xj: the Float64 loops (w promoted like `w .* xj`)."""
    s = _i64(s)
    t = _i64(t)
    if _is64(xj):
        xj = _f64(xj)
        w = None if w is None else _f64(w)
        n_dst = n if n_dst is None else n_dst
        D = int(np.prod(xj.shape[1:], dtype=np.int64))
        out = np.empty((n_dst,) + xj.shape[1:], np.float64)
        _check(lib().orc_propagate64(_AGGR[aggr], _p(s), _p(t), _i(s.shape[0]), _i(xj.shape[0]), _i(n_dst), _p(xj), _i(D), _p(w), _p(out)),
               "propagate")
        return out
    xj = _f32(xj)
    w = None if w is None else _f32(w)
    n_dst = n if n_dst is None else n_dst
    D = int(np.prod(xj.shape[1:], dtype=np.int64))
    out = np.empty((n_dst,) + xj.shape[1:], np.float32)
    _check(lib().orc_propagate(_AGGR[aggr], _p(s), _p(t), _i(s.shape[0]), _i(xj.shape[0]), _i(n_dst), _p(xj),
                               _i(D), _p(w), _p(out)), "propagate")
    return out


def spmm_csc(s, t, n, x, w=None):
    """CPU fast path `xj * adjacency_matrix(g)` — GNNlib/src/msgpass.jl:215-238, convert.jl:221-237."""
    s = _i64(s)
    t = _i64(t)
    x = _f32(x)
    w = None if w is None else _f32(w)
    out = np.empty_like(x)
    _check(lib().orc_spmm_csc(_p(s), _p(t), _p(w), _i(s.shape[0]), _i(n), _p(x), _i(x.shape[1]), _p(out)), "spmm_csc")
    return out


def softmax_edge_neighbors(t, n, e):
    """softmax_edge_neighbors(g, e) — GNNlib/src/utils.jl:84-97.  e [E, H] (or [E])."""
    t = _i64(t)
    e = _f32(e)
    H = int(np.prod(e.shape[1:], dtype=np.int64))
    out = np.empty_like(e)
    _check(lib().orc_softmax_edge_neighbors(_p(t), _i(t.shape[0]), _i(n), _p(e), _i(H), _p(out)), "softmax_edge_neighbors")
    return out


def reduce_nodes(aggr, graph_indicator, x, num_graphs=None):
    """reduce_nodes(aggr, g, x) = NNlib.scatter(aggr, x, graph_indicator) — GNNlib/src/utils.jl:12-16."""
    return scatter(aggr, x, graph_indicator, num_graphs)


def scale_rows(x, c):
    x = _f32(x)
    c = _f32(c)
    out = np.empty_like(x)
    lib().orc_scale_rows(_p(x), _p(c), _i(x.shape[0]), _i(x.shape[1]), _p(out))
    return out


def inv_sqrt(d):
    d = _f32(d)
    out = np.empty_like(d)
    lib().orc_inv_sqrt(_p(d), _i(d.shape[0]), _p(out))
    return out


def matmul(W, x, blas=True):
    """`weight * x` for weight (Dout, Din), x [N, Din] -> [N, Dout].  blas=True uses numpy/OpenBLAS sgemm (what the
    reference does); blas=False the plain k-ordered C loop."""
    W = _f32(W)
    x = _f32(x)
    if blas:
        return _f32(x @ W.T)
    y = np.empty((x.shape[0], W.shape[0]), np.float32)
    lib().orc_matmul(_p(W), _p(x), _i(x.shape[0]), _i(W.shape[0]), _i(W.shape[1]), _p(y))
    return y


def _act(sigma, x):
    if sigma in (None, "identity"):
        return x
    if sigma == "relu":
        return np.where(x < 0, np.float32(0), x).astype(np.float32)  # NNlib.relu = ifelse(x < 0, zero(x), x)
    if sigma == "tanh":
        return np.tanh(x).astype(np.float32)
    raise ValueError(sigma)


# ---------------------------------------------------------------------------------------------------------
# layer bodies (GNNlib/src/layers/conv.jl, pool.jl)
# ---------------------------------------------------------------------------------------------------------
def gcn_conv(s, t, n, x, weight, bias=None, sigma=None, add_self_loops_=True, use_edge_weight=False,
             graph_w=None, edge_weight=None, fast_path=False, blas=True):
    """gcn_conv — GNNlib/src/layers/conv.jl:14-72.  graph_w: the graph's own weights (used iff use_edge_weight);
    edge_weight: the optional call argument.  fast_path selects the CPU SpMM specialisation (msgpass.jl:215-238)
    instead of the generic gather/scatter."""
    s = _i64(s)
    t = _i64(t)
    x = _f32(x)
    weight = _f32(weight)
    if edge_weight is not None and len(edge_weight) != len(s):
        raise ValueError("Wrong number of edge weights")  # ArgumentError, conv.jl:6-10
    gw = graph_w
    if add_self_loops_:
        s, t, gw = add_self_loops(s, t, n, gw)
        if edge_weight is not None:
            edge_weight = np.concatenate([_f32(edge_weight), np.ones(n, np.float32)])  # conv.jl:28-33
    Dout, Din = weight.shape
    if Dout < Din:
        x = matmul(weight, x, blas)
    if edge_weight is not None:
        d = degree(t, n, edge_weight)
    else:
        d = degree(t, n, gw if use_edge_weight else None)
    c = inv_sqrt(d)
    xj = scale_rows(x, c)
    if edge_weight is not None:
        wmsg = edge_weight
    elif use_edge_weight:
        wmsg = gw
    else:
        wmsg = None
    if fast_path:
        x = spmm_csc(s, t, n, xj, wmsg)
    else:
        x = propagate(SUM, s, t, n, xj, wmsg)
    x = scale_rows(x, c)
    if Dout >= Din:
        x = matmul(weight, x, blas)
    if bias is not None:
        x = x + _f32(bias)[None, :]
    return _act(sigma, x)


def _propagate_copy_xj(aggr, s, t, n, x, fast_path):
    """propagate(copy_xj, g, aggr; xj = x) as the reference dispatches it on CPU arrays: `+` hits the SpMM specialisation
    `xj * adjacency_matrix(g, weighted = false)` (msgpass.jl:215-218), every other aggr the generic path."""
    if fast_path and _AGGR[aggr] == SUM:
        return spmm_csc(s, t, n, x)
    return propagate(aggr, s, t, n, x)


def graph_conv(s, t, n, x, weight1, weight2, bias=None, sigma=None, aggr=SUM, blas=True, fast_path=False):
    """graph_conv — GNNlib/src/layers/conv.jl:102-108."""
    x = _f32(x)
    m = _propagate_copy_xj(aggr, s, t, n, x, fast_path)
    y = matmul(weight1, x, blas) + matmul(weight2, m, blas)
    if bias is not None:
        y = y + _f32(bias)[None, :]
    return _act(sigma, y)


def sage_conv(s, t, n, x, weight, bias=None, sigma=None, aggr=MEAN, blas=True, fast_path=False):
    """sage_conv — GNNlib/src/layers/conv.jl:277-283.  weight (Dout, 2*Din)."""
    x = _f32(x)
    m = _propagate_copy_xj(aggr, s, t, n, x, fast_path)
    y = matmul(weight, np.concatenate([x, m], axis=1), blas)
    if bias is not None:
        y = y + _f32(bias)[None, :]
    return _act(sigma, y)


def dropout_keep(seed, p, n_edges, heads):
    """The keep mask of the attention dropout as include/gnnmp.h (gnnmp_gat_conv_drop_f32) defines it — the build's own counter-based
    generator, restated with numpy uint32 arithmetic (the reference draws its mask from Julia's task-local RNG: only the distribution,
    Bernoulli(1 - p) per coefficient, can be matched, never the draws).  keep[e, h] in {0, 1}, e = 0-based edge position."""
    def mix(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
        return x
    lo, hi = np.uint32(seed & 0xffffffff), np.uint32((seed >> 32) & 0xffffffff)
    e = np.arange(n_edges, dtype=np.uint32)[:, None]
    h = np.arange(heads, dtype=np.uint32)[None, :]
    with np.errstate(over="ignore"):
        bits = mix(mix(e ^ lo) ^ (h * np.uint32(0x9e3779b9) + hi))
    thr = min(int(float(np.float32(p)) * 4294967296.0), 0xffffffff)
    return (bits >= np.uint32(thr)).astype(np.uint8)


def gat_conv(s, t, n, x, dense_x_weight, a, bias=None, sigma=None, heads=1, concat=True, negative_slope=0.2,
             add_self_loops_=True, blas=True, return_alpha=False, dropout=0.0, seed=0):
    """gat_conv + gat_message — GNNlib/src/layers/conv.jl:112-167.
    dense_x_weight (C*H, Din); a (2C, H) in Julia shape, i.e. numpy [2C, H].
    dropout > 0: α = dropout(α, p) (conv.jl:139; NNlib.dropout: α .* keep ./ (1 - p)) with the mask of dropout_keep(seed, ...)."""
    s = _i64(s)
    t = _i64(t)
    x = _f32(x)
    a = _f32(a)
    H = heads
    C = dense_x_weight.shape[0] // H
    if add_self_loops_:
        s, t, _ = add_self_loops(s, t, n)
    Wx = matmul(dense_x_weight, x, blas).reshape(n, H, C)       # reshape(dense_x(x), C, H, N)
    Wxi = gather(Wx, t)                                          # msgpass.jl:125
    Wxj = gather(Wx, s)                                          # msgpass.jl:126
    a_hc = _f32(a.T)                                             # [H][2C]
    E = s.shape[0]
    logit = np.empty((E, H), np.float32)
    lib().orc_gat_logits(_p(Wxi), _p(Wxj), _p(a_hc), _i(E), _i(H), _i(C), ctypes.c_float(negative_slope), _p(logit))
    alpha = softmax_edge_neighbors(t, n, logit)                  # utils.jl:84-97
    if dropout > 0.0:                                            # conv.jl:139
        keep = dropout_keep(seed, dropout, E, H).astype(np.float32)
        alpha = _f32(alpha * keep * np.float32(1.0 / (1.0 - np.float32(dropout))))
    beta = np.empty_like(Wxj)
    lib().orc_gat_weight_messages(_p(alpha), _p(Wxj), _i(E), _i(H), _i(C), _p(beta))
    y = scatter(SUM, beta.reshape(E, H * C), t, n).reshape(n, H, C)   # aggregate_neighbors(g, +, β)
    if not concat:
        # mean(x, dims = 2): sum over heads then divide
        acc = y[:, 0, :].copy()
        for h in range(1, H):
            acc = acc + y[:, h, :]
        y = acc / np.float32(H)
    y = y.reshape(n, -1)
    if bias is not None:
        y = y + _f32(bias)[None, :]
    y = _act(sigma, y)
    return (y, alpha) if return_alpha else y


def global_pool(aggr, graph_indicator, x, num_graphs=None):
    """global_pool — GNNlib/src/layers/pool.jl:3-5."""
    return reduce_nodes(aggr, graph_indicator, x, num_graphs)


# ---------------------------------------------------------------------------------------------------------
# adjoints — NNlib's rrules restated (what Zygote runs through the generic path)
#   ∇gather(x, idx): Δx = scatter(+, Δ, idx)                     ∇scatter(+): Δsrc = gather(Δ, idx)
#   ∇scatter(mean):  Δsrc = gather(Δ, idx) ./ count[idx]          ∇scatter(max|min): (src .== gather(dst, idx)) .* gather(Δ, idx)
# ---------------------------------------------------------------------------------------------------------
def grad_propagate(aggr, s, t, n, dy, xj, w=None):
    """(Δxj, Δw) of propagate(copy_xj | w_mul_xj, g, aggr; xj [, w]) for an incoming Δ = dy [n, D]"""
    s = _i64(s)
    t = _i64(t)
    dy = _f32(dy)
    xj = _f32(xj)
    code = _AGGR[aggr]
    dm = gather(dy, t)                                       # ∇scatter(+)
    m = gather(xj, s)
    if w is not None:
        m = (_f32(w)[:, None] * m).astype(np.float32)
    if code == MEAN:
        cnt = np.bincount(t - 1, minlength=n).astype(np.float32)
        dm = (dm / cnt[t - 1][:, None]).astype(np.float32)
    elif code in (MAX, MIN):
        y = scatter(aggr, m, t, n)
        dm = ((m == gather(y, t)) * dm).astype(np.float32)
    dw = None
    if w is not None:
        dw = (dm * gather(xj, s)).sum(axis=1).astype(np.float32)   # sum(Δm .* xj, dims = 1)
        dm = (_f32(w)[:, None] * dm).astype(np.float32)
    dx = scatter(SUM, dm, s, n)                              # ∇gather
    return dx, dw


def grad_gcn_conv(s, t, n, x, weight, bias, sigma, dy, add_self_loops_=True):
    """(Δx, ΔW, Δb) of gcn_conv (default norm, unweighted), composed from the rules above + the dense rules
    (Δz = Δy .* σ'(z), ΔW = Δz' * a, Δb = sum(Δz), Δa = Δz * W), float32 with float64 accumulation in the matmuls."""
    s = _i64(s)
    t = _i64(t)
    x = _f32(x)
    W = _f32(weight)
    if add_self_loops_:
        s, t, _ = add_self_loops(s, t, n)
    c = inv_sqrt(degree(t, n))
    Dout, Din = W.shape

    def P(h):
        return scale_rows(propagate(SUM, s, t, n, scale_rows(h, c)), c)

    def PT(dh):
        dxs, _ = grad_propagate(SUM, s, t, n, scale_rows(dh, c), np.zeros((n, dh.shape[1]), np.float32))
        return scale_rows(dxs, c)

    if Dout < Din:
        h = matmul(W, x)
        z = P(h) + (0 if bias is None else _f32(bias)[None, :])
        dz = _f32(dy) * (z > 0) if sigma == "relu" else _f32(dy)
        db = dz.astype(np.float64).sum(0).astype(np.float32)
        dh = PT(dz.astype(np.float32))
        dW = (dh.astype(np.float64).T @ x.astype(np.float64)).astype(np.float32)
        dx = (dh.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)
    else:
        a = P(x)
        z = matmul(W, a) + (0 if bias is None else _f32(bias)[None, :])
        dz = (_f32(dy) * (z > 0)).astype(np.float32) if sigma == "relu" else _f32(dy)
        db = dz.astype(np.float64).sum(0).astype(np.float32)
        dW = (dz.astype(np.float64).T @ a.astype(np.float64)).astype(np.float32)
        dx = PT((dz.astype(np.float64) @ W.astype(np.float64)).astype(np.float32))
    return dx, dW, db


def grad_gat_conv(s, t, n, x, dense_x_weight, a, bias, sigma, dy, heads=1, negative_slope=0.2, add_self_loops_=True,
                  concat=True, dropout=0.0, seed=0):
    """(Δx, ΔW, Δa, Δb) of gat_conv (concat = true, no edge features; conv.jl:112-167), composed rule by rule in the order
    Zygote walks the forward backwards: σ, bias, ∇scatter(+) (Δβ = Δ[t]), β = α .* Wxj, the softmax_edge_neighbors
    pullback  Δl = α .* (Δα - Σ_{N(i)} α Δα)  (utils.jl:84-97 is exp / scatter / gather / division, this is what their
    rules multiply out to), leakyrelu', the `sum(a .* vcat(Wxi, Wxj))` contraction, ∇gather (scatter(+) to t and to s) and
    the dense rules.  float64 arithmetic on the float32 inputs, results cast to float32.  a: Julia shape (2C, H)."""
    s = _i64(s)
    t = _i64(t)
    if add_self_loops_:
        s, t, _ = add_self_loops(s, t, n)
    si, ti = s - 1, t - 1
    H = heads
    W = np.asarray(dense_x_weight, np.float64)
    C = W.shape[0] // H
    x64 = np.asarray(x, np.float64)
    a64 = np.asarray(a, np.float64)
    ad, as_ = a64[:C].T, a64[C:].T                                # [H, C] target / source halves
    Wx = (x64 @ W.T).reshape(n, H, C)
    Wxi, Wxj = Wx[ti], Wx[si]
    z = (Wxi * ad).sum(-1) + (Wxj * as_).sum(-1)                  # [E', H]
    l = np.where(z > 0, z, negative_slope * z)
    m = np.full((n, H), -np.inf)
    np.maximum.at(m, ti, l)
    p = np.exp(l - m[ti])
    den = np.zeros((n, H))
    np.add.at(den, ti, p)
    alpha = p / den[ti]
    # α' = dropout(α) = k .* α with the constant k = keep / (1 - p) (conv.jl:139): its rule hands Δα = k .* Δα' to the softmax pullback
    k = dropout_keep(seed, dropout, len(ti), H).astype(np.float64) / (1.0 - float(np.float32(dropout))) if dropout > 0.0 else 1.0
    o = np.zeros((n, H, C))
    np.add.at(o, ti, (alpha * k)[..., None] * Wxj)
    y = (o.reshape(n, H * C) if concat else o.mean(axis=1)) + (0 if bias is None else np.asarray(bias, np.float64)[None, :])
    dz = np.asarray(dy, np.float64) * (y > 0) if sigma == "relu" else np.asarray(dy, np.float64)
    db = dz.sum(0)
    delta = dz.reshape(n, H, C) if concat else np.repeat(dz[:, None, :] / H, H, axis=1)   # ∇mean(x, dims = 2)
    dbeta = delta[ti]                                             # ∇scatter(+)
    dalpha = (dbeta * Wxj).sum(-1) * k
    dWxj = (alpha * k)[..., None] * dbeta
    sa = np.zeros((n, H))
    np.add.at(sa, ti, alpha * dalpha)
    dzl = alpha * (dalpha - sa[ti]) * np.where(z > 0, 1.0, negative_slope)
    dWxi = dzl[..., None] * ad[None]
    dWxj = dWxj + dzl[..., None] * as_[None]
    da_d = (dzl[..., None] * Wxi).sum(0)
    da_s = (dzl[..., None] * Wxj).sum(0)
    dWx = np.zeros((n, H, C))
    np.add.at(dWx, ti, dWxi)                                      # ∇gather(Wx, t)
    np.add.at(dWx, si, dWxj)                                      # ∇gather(Wx, s)
    dWx = dWx.reshape(n, H * C)
    dW = dWx.T @ x64
    dx = dWx @ W
    da = np.concatenate([da_d.T, da_s.T], axis=0)                 # (2C, H)
    f = np.float32
    return dx.astype(f), dW.astype(f), da.astype(f), db.astype(f)


def grad_graph_conv(s, t, n, x, weight1, weight2, bias, sigma, dy, aggr=SUM):
    """(Δx, ΔW1, ΔW2, Δb) of graph_conv (conv.jl:102-108); sage_conv (conv.jl:277-283) is the same with W = [W1 W2].
    Dense rules in float64, the propagate pullback through grad_propagate above."""
    x = _f32(x)
    W1 = np.asarray(weight1, np.float64)
    W2 = np.asarray(weight2, np.float64)
    m = propagate(aggr, s, t, n, x)
    z = x.astype(np.float64) @ W1.T + m.astype(np.float64) @ W2.T + (0 if bias is None else np.asarray(bias, np.float64)[None, :])
    dz = np.asarray(dy, np.float64) * (z > 0) if sigma == "relu" else np.asarray(dy, np.float64)
    db = dz.sum(0)
    dW1 = dz.T @ x.astype(np.float64)
    dW2 = dz.T @ m.astype(np.float64)
    dm = (dz @ W2).astype(np.float32)
    dxm, _ = grad_propagate(aggr, s, t, n, dm, x)
    dx = dz @ W1 + dxm.astype(np.float64)
    f = np.float32
    return dx.astype(f), dW1.astype(f), dW2.astype(f), db.astype(f)
