"""Graph-wise and neighbourhood-wise reductions built on the engine.

Mirror of GNNlib/src/utils.jl:
  reduce_nodes(aggr, g, x) / reduce_nodes(aggr, indicator, x)     utils.jl:12-28
  softmax_edge_neighbors(g, e)                                   utils.jl:84-97
  expand_srcdst                                                  utils.jl:123-125
and GNNlib/src/layers/pool.jl:3-5 (global_pool).
"""
from __future__ import annotations

import torch

from . import _lib as L
from .graph import GNNGraph, graph_indicator
from .msgpass import _flat, _idx_plan, _scatter_plan, aggr_code


def reduce_nodes(aggr, g, x, num_graphs=None, sorted_indicator=None):
    """reduce_nodes(aggr, g::GNNGraph, x) = NNlib.scatter(aggr, x, graph_indicator(g)) — utils.jl:12-16; the second
    form takes the indicator vector directly (utils.jl:26-28).  A `batch`-built indicator is sorted, so graphs are
    contiguous segments (gnnmp_segment_pool_f32); an arbitrary indicator goes through a plan (scatter in node order)."""
    if isinstance(g, GNNGraph):
        assert x.shape[0] == g.num_nodes
        gi = graph_indicator(g)
        G = g.num_graphs
        base = g.index_base
        is_sorted = True  # MLUtils.batch builds fill(1,n1); fill(2,n2); ... (transform.jl:691-699)
    else:
        gi = g
        base = 1
        # NNlib.scatter without dstsize sizes the output by maximum(idx)
        G = int(gi.max()) - base + 1 if num_graphs is None else num_graphs
        if sorted_indicator is not None:
            is_sorted = bool(sorted_indicator)
        else:
            import ctypes
            res = ctypes.c_int(0)
            L.check(L.load().gnnmp_is_sorted(L.ptr(gi), 8 if gi.dtype == torch.int64 else 4, gi.numel(), ctypes.byref(res),
                                             L.stream_ptr()))
            is_sorted = bool(res.value)
    xf = _flat(x)
    if not is_sorted:
        return _scatter_plan(aggr, x, _idx_plan(gi, G, base))
    out = torch.empty((G,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    L.check(L.load().gnnmp_segment_pool_f32(aggr_code(aggr), L.ptr(xf), L.ptr(gi), 8 if gi.dtype == torch.int64 else 4,
                                            base, L.ptr(out), xf.shape[1], xf.shape[0], G, L.stream_ptr()))
    return out


def softmax_edge_neighbors(g: GNNGraph, e):
    """Softmax over each node's incoming edges — utils.jl:84-97.  e: [num_edges, ...]"""
    assert e.shape[0] == g.num_edges
    ef = _flat(e)
    out = torch.empty_like(ef)
    L.check(L.load().gnnmp_edge_softmax_f32(g.plan(False).handle, L.ptr(ef), L.ptr(out), ef.shape[1], L.stream_ptr()))
    return out.view(e.shape)


def expand_srcdst(g, x):
    """utils.jl:123-125"""
    if isinstance(x, torch.Tensor) and x.dim() == 2:
        return x, x
    if isinstance(x, tuple) and len(x) == 2:
        return x
    raise ValueError("Invalid input type, expected matrix or tuple of matrices.")
