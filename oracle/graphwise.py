"""CPU restatement of the graph-wise helpers of GNNlib/src/utils.jl — TEST INFRASTRUCTURE ONLY (same rules as oracle.py).

    reduce_edges     utils.jl:37-42      softmax_nodes   utils.jl:49-57      softmax_edges    utils.jl:64-72
    broadcast_nodes  utils.jl:104-108    broadcast_edges utils.jl:116-120
Each is the reference's own composition of gather / scatter (pinned in oracle.py), float32, statement by statement.
Pinned by the properties the reference's tests assert (GNNlib/test/utils.jl:22-56): tests/test_graphwise.py.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

f32 = np.float32


def edge_indicator(graph_indicator, s):
    """graph_indicator(g, edges = true) = graph_indicator(g)[s] — GNNGraphs/src/query.jl:507-509"""
    return O._i64(graph_indicator)[O._i64(s) - 1]


def reduce_edges(aggr, graph_indicator, s, e, num_graphs):
    return O.scatter(aggr, O._f32(e), edge_indicator(graph_indicator, s), num_graphs)


def _segment_softmax(x, gi, G, den_add):
    x = O._f32(x)
    gi = O._i64(gi)
    max_ = O.gather(O.scatter(O.MAX, x, gi, G), gi)
    num = np.exp((x - max_).astype(f32)).astype(f32)
    den = O.gather(O.scatter(O.SUM, num, gi, G), gi)
    if den_add:
        den = (den + f32(den_add)).astype(f32)
    return (num / den).astype(f32)


def softmax_nodes(graph_indicator, x, num_graphs):
    return _segment_softmax(x, graph_indicator, num_graphs, 0.0)


def softmax_edges(graph_indicator, s, e, num_graphs):
    return _segment_softmax(e, edge_indicator(graph_indicator, s), num_graphs, np.finfo(np.float32).eps)


def broadcast_nodes(graph_indicator, x):
    return O.gather(O._f32(x), O._i64(graph_indicator))


def broadcast_edges(graph_indicator, s, x):
    return O.gather(O._f32(x), edge_indicator(graph_indicator, s))
