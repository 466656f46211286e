"""Message passing: propagate / apply_edges / aggregate_neighbors and the built-in message functions.

Mirror of GNNlib/src/msgpass.jl (same names, argument order, error behaviour):
  propagate(f, g, aggr; xi, xj, e)         msgpass.jl:71-79   (+ fast-path specialisations :215-238)
  apply_edges(f, g; xi, xj, e)             msgpass.jl:121-129
  aggregate_neighbors(g, aggr, m)          msgpass.jl:145-149
  copy_xj, copy_xi, xi_dot_xj, xi_sub_xj, xj_sub_xi, e_mul_xj, w_mul_xj      msgpass.jl:162-208
and of the leaf ops GNNGraphs/src/gatherscatter.jl:1-18 (`_gather`, `_scatter`, recursing over tuples / dicts / None).

Where the reference's GPU extension (GNNlib/ext/GNNlibAMDGPUExt.jl:13-32) routes `copy_xj / e_mul_xj / w_mul_xj` to the
generic three-pass gather -> message -> atomic-scatter path, this module calls ONE fused HIP kernel
(`gnnmp_propagate_f32`) for every aggr in {+, mean, max, min}.  Arbitrary Python closures still work: they take the
generic path built from the HIP gather and the deterministic plan-based HIP scatter.
"""
from __future__ import annotations

import operator
import statistics

import torch

from . import _lib as L
from .graph import GNNGraph, Plan, check_num_edges, check_num_nodes, edge_index

_AGGR = {
    "+": L.SUM, "sum": L.SUM, "add": L.SUM, operator.add: L.SUM, sum: L.SUM, torch.sum: L.SUM, torch.add: L.SUM,
    "mean": L.MEAN, statistics.mean: L.MEAN, torch.mean: L.MEAN,
    "max": L.MAX, max: L.MAX, torch.max: L.MAX, torch.maximum: L.MAX,
    "min": L.MIN, min: L.MIN, torch.min: L.MIN, torch.minimum: L.MIN,
}


def aggr_code(aggr) -> int:
    try:
        return _AGGR[aggr]
    except (KeyError, TypeError):
        raise ValueError(f"unsupported aggregation operator {aggr!r} (expected +, mean, max or min)")


def _flat(x: torch.Tensor):
    """[N, ...] -> contiguous float32 [N, D]"""
    assert x.dtype == torch.float32, "gnnmp supports Float32 features (Float64: propagate / _gather / _scatter only)"
    x = x.contiguous()
    return x.view(x.shape[0], -1) if x.dim() != 2 else x


def _flat_any(x: torch.Tensor):
    """[N, ...] -> contiguous [N, D] of x's own eltype — Float32 or Float64 (the *_f64 entry points: gnnmp.h, round 6)"""
    assert x.dtype in (torch.float32, torch.float64), "gnnmp supports Float32 and (for propagate / _gather / _scatter) Float64 features"
    x = x.contiguous()
    return x.view(x.shape[0], -1) if x.dim() != 2 else x


def _sfx(t: torch.Tensor):
    return "f64" if t.dtype == torch.float64 else "f32"


# ---------------------------------------------------------------------------------------------------------
# leaf ops
# ---------------------------------------------------------------------------------------------------------
def _gather(x, i, index_base=1):
    """GNNGraphs/src/gatherscatter.jl:1-5 — NNlib.gather on the last (here: first) dimension."""
    if x is None:
        return None
    if isinstance(x, dict):
        return {k: _gather(v, i, index_base) for k, v in x.items()}
    if isinstance(x, tuple):
        return tuple(_gather(v, i, index_base) for v in x)
    if isinstance(x, list):
        return [_gather(v, i, index_base) for v in x]
    xf = _flat_any(x)
    K, D = i.numel(), xf.shape[1]
    out = torch.empty((K,) + tuple(x.shape[1:]), dtype=xf.dtype, device=x.device)
    L.check(getattr(L.load(), "gnnmp_gather_" + _sfx(xf))(L.ptr(xf), L.ptr(i), 8 if i.dtype == torch.int64 else 4, index_base, K,
                                                          L.ptr(out), D, L.stream_ptr()))
    return out


def _scatter_plan(aggr, src, plan: Plan):
    """GNNGraphs/src/gatherscatter.jl:7-18 with idx = the plan's targets: deterministic, edge-order reduction."""
    if src is None:
        return None
    if isinstance(src, dict):
        return {k: _scatter_plan(aggr, v, plan) for k, v in src.items()}
    if isinstance(src, tuple):
        return tuple(_scatter_plan(aggr, v, plan) for v in src)
    if isinstance(src, list):
        return [_scatter_plan(aggr, v, plan) for v in src]
    sf = _flat_any(src)
    assert sf.shape[0] == plan.n_total
    out = torch.empty((plan.n_dst,) + tuple(src.shape[1:]), dtype=sf.dtype, device=src.device)
    L.check(getattr(L.load(), "gnnmp_scatter_" + _sfx(sf))(plan.handle, aggr_code(aggr), L.ptr(sf), L.ptr(out), sf.shape[1],
                                                           L.stream_ptr()))
    return out


def _scatter(aggr, src, idx, n, index_base=1):
    """NNlib.scatter(aggr, src, idx; dstsize = (..., n)) for an arbitrary index vector: builds a throw-away plan
    (src = 1..K -> dst = idx) so that the reduction is in k order, like NNlib's CPU loop."""
    if src is None:
        return None
    return _scatter_plan(aggr, src, _idx_plan(idx, n, index_base))


def _idx_plan(idx, n, index_base):
    K = idx.numel()
    ar = torch.arange(index_base, K + index_base, dtype=idx.dtype, device=idx.device)
    return Plan(ar, idx, K, n, index_base, False, validate=True)


# ---------------------------------------------------------------------------------------------------------
# message functions — GNNlib/src/msgpass.jl:162-208
# ---------------------------------------------------------------------------------------------------------
def copy_xj(xi, xj, e):
    return xj


def copy_xi(xi, xj, e):
    return xi


def xi_dot_xj(xi, xj, e):
    return (xi * xj).sum(dim=-1, keepdim=True)  # Julia dims = 1 is the fastest (here: last) feature dim


def xi_sub_xj(xi, xj, e):
    return xi - xj


def xj_sub_xi(xi, xj, e):
    return xj - xi


def e_mul_xj(xi, xj, e):
    assert e.dim() <= xj.dim()
    # Julia reshapes e to (1,...,1, size(e)...) i.e. prepends singleton FEATURE dims; row-major: append them
    # between the edge dim and the trailing feature dims that e does not have.
    shape = (e.shape[0],) + (1,) * (xj.dim() - e.dim()) + tuple(e.shape[1:])
    return e.reshape(shape) * xj


def w_mul_xj(xi, xj, w):
    if w is None:
        return xj
    return w.reshape((w.shape[0],) + (1,) * (xj.dim() - 1)) * xj


# ---------------------------------------------------------------------------------------------------------
# apply_edges / aggregate_neighbors / propagate
# ---------------------------------------------------------------------------------------------------------
def apply_edges(f, g: GNNGraph, xi=None, xj=None, e=None):
    """msgpass.jl:121-129"""
    check_num_nodes(g, (xj, xi))
    check_num_edges(g, e)
    s, t = edge_index(g)
    if isinstance(xi, torch.Tensor) and isinstance(xj, torch.Tensor) and xi.shape[1:] == xj.shape[1:] \
            and xi.dtype == torch.float32 and xj.dtype == torch.float32:      # (Float64: the generic gather -> f -> scatter path below)
        # the two-row message functions in one pass over the edges (no gathered (D, E) temporaries)
        if f is xi_dot_xj and xi.dim() == 2:
            out = torch.empty((g.num_edges, 1), dtype=torch.float32, device=xi.device)
            if g.num_edges > 0 and g.num_nodes > 0:
                # walked in the plan's destination-sorted order: the row of xi stays in registers for all of a destination's edges — half
                # the gathered bytes of the COO-order kernel (products shape: 6.6 against 10.3 ms); results land in original edge order.
                # Rows wider than a wave of 16-byte lanes (D > 256): GNNMP_EUNSUPPORTED, the COO-order kernel below.
                rc = L.load().gnnmp_edge_dot_plan_f32(g.plan(False).handle, L.ptr(xi.contiguous()), L.ptr(xj.contiguous()), L.ptr(out),
                                                      xi.shape[1], L.stream_ptr())
                if rc == L.OK:
                    return out
                if rc != L.EUNSUPPORTED:
                    L.check(rc)
            L.check(L.load().gnnmp_edge_dot_f32(L.ptr(xi.contiguous()), L.ptr(xj.contiguous()), L.ptr(s), L.ptr(t),
                                                g.idx_bytes, g.index_base, g.num_edges, xi.shape[1], L.ptr(out),
                                                L.stream_ptr()))
            return out
        if f is xi_sub_xj or f is xj_sub_xi:
            xif, xjf = _flat(xi), _flat(xj)
            out = torch.empty((g.num_edges,) + tuple(xi.shape[1:]), dtype=torch.float32, device=xi.device)
            L.check(L.load().gnnmp_edge_sub_f32(L.ptr(xif), L.ptr(xjf), L.ptr(s), L.ptr(t), g.idx_bytes, g.index_base,
                                                g.num_edges, 1 if f is xj_sub_xi else 0, L.ptr(out), xif.shape[1],
                                                L.stream_ptr()))
            return out
    xi = _gather(xi, t, g.index_base)
    xj = _gather(xj, s, g.index_base)
    return f(xi, xj, e)


def aggregate_neighbors(g: GNNGraph, aggr, m):
    """msgpass.jl:145-149"""
    check_num_edges(g, m)
    return _scatter_plan(aggr, m, g.plan(False))


def _fused(g: GNNGraph, msg: int, aggr, xj, w, scale_src=None, scale_dst=None, add_self_loops=False, out=None):
    plan = g.plan(add_self_loops)
    xf = _flat_any(xj)
    if out is None:
        out = torch.empty((plan.n_dst,) + tuple(xj.shape[1:]), dtype=xf.dtype, device=xj.device)
    else:
        assert out.shape == (plan.n_dst,) + tuple(xj.shape[1:]) and out.dtype == xf.dtype and out.is_contiguous()
    # Float64 features (the reference's own micro-benchmark, perf/bench_gnn.jl:9-10): weights and factors in the features' eltype
    conv = (lambda t: None if t is None else t.to(xf.dtype).contiguous())
    w, scale_src, scale_dst = conv(w), conv(scale_src), conv(scale_dst)
    L.check(getattr(L.load(), "gnnmp_propagate_" + _sfx(xf))(plan.handle, msg, aggr_code(aggr), L.ptr(xf), L.ptr(w), L.ptr(scale_src),
                                                             L.ptr(scale_dst), L.ptr(out), xf.shape[1], L.stream_ptr()))
    return out


def propagate(f, g: GNNGraph, aggr, xi=None, xj=None, e=None):
    """msgpass.jl:71-79.  `copy_xj`, `w_mul_xj` (graph weights) and `e_mul_xj` with a vector `e` take the fused kernel
    for every aggr (the reference specialises only `+`, and only on the CPU: msgpass.jl:215-238)."""
    if isinstance(xj, torch.Tensor):
        if f is copy_xj:
            check_num_nodes(g, (xj, xi))
            check_num_edges(g, e)
            return _fused(g, L.COPY_XJ, aggr, xj, None)
        if f is w_mul_xj and e is None:
            check_num_nodes(g, (xj, xi))
            if g.w is None:
                return _fused(g, L.COPY_XJ, aggr, xj, None)
            return _fused(g, L.W_MUL_XJ, aggr, xj, g.w)
        if f is e_mul_xj and isinstance(e, torch.Tensor) and e.dim() == 1:
            check_num_nodes(g, (xj, xi))
            check_num_edges(g, e)
            if e.dtype == torch.float64 and xj.dtype == torch.float32:      # Julia's `e .* xj` promotes: Float64 result
                xj = xj.to(torch.float64)
            return _fused(g, L.W_MUL_XJ, aggr, xj, e)      # (e converted to xj's eltype in _fused)
        if f is e_mul_xj and isinstance(e, torch.Tensor) and e.dim() == xj.dim() and e.shape[1:] == xj.shape[1:] \
                and e.dtype == torch.float32:
            # matrix e (D, E): one row of factors per edge, fetched by edge id inside the fused kernel
            check_num_nodes(g, (xj, xi))
            check_num_edges(g, e)
            xf, ef = _flat(xj), _flat(e)
            plan = g.plan(False)
            out = torch.empty((plan.n_dst,) + tuple(xj.shape[1:]), dtype=torch.float32, device=xj.device)
            L.check(L.load().gnnmp_propagate_emul_f32(plan.handle, aggr_code(aggr), L.ptr(xf), L.ptr(ef), L.ptr(out),
                                                      xf.shape[1], L.stream_ptr()))
            return out
    m = apply_edges(f, g, xi, xj, e)
    return aggregate_neighbors(g, aggr, m)
