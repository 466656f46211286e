"""Device-side graph preparation next to the path: sort_edge_index, is_bidirected,
has_self_loops (GNNGraphs/src/utils.jl:30-45, query.jl:553-569) and sample_neighbors (GNNGraphs/src/sampling.jl:68-119).
Index work only; every step is a libgnnmp call (csrc/graphprep.hip) — torch allocates.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L
from .graph import GNNGraph, Plan


def _ib(v):
    return 8 if v.dtype == torch.int64 else 4


def sort_edge_index(u, v=None, index_base=1):
    """sort_edge_index(u, v) / sort_edge_index((u, v)) -> (u', v'): pairs sorted lexicographically — utils.jl:30-45"""
    if v is None:
        u, v = u
    assert u.dtype == v.dtype and u.dtype in (torch.int64, torch.int32) and u.shape == v.shape and u.dim() == 1
    u, v = u.contiguous(), v.contiguous()
    uo, vo = torch.empty_like(u), torch.empty_like(v)
    L.check(L.load().gnnmp_sort_edge_index(L.ptr(u), L.ptr(v), _ib(u), index_base, u.numel(), L.ptr(uo), L.ptr(vo),
                                           L.stream_ptr()))
    return uo, vo


def is_bidirected(g: GNNGraph) -> bool:
    """query.jl:553-558"""
    res = ctypes.c_int(0)
    L.check(L.load().gnnmp_is_bidirected(L.ptr(g.s), L.ptr(g.t), g.idx_bytes, g.index_base, g.num_edges, ctypes.byref(res),
                                         L.stream_ptr()))
    return bool(res.value)


def has_self_loops(g: GNNGraph) -> bool:
    """query.jl:565-569"""
    res = ctypes.c_int(0)
    L.check(L.load().gnnmp_has_self_loops(L.ptr(g.s), L.ptr(g.t), g.idx_bytes, g.num_edges, ctypes.byref(res),
                                          L.stream_ptr()))
    return bool(res.value)


def _take_index(v, pos, index_base):
    """v[pos] for an index vector v (bit copy through the float gather: one or two 32-bit words per element)"""
    words = 2 if v.dtype == torch.int64 else 1
    src = v.contiguous().view(torch.float32).view(v.numel(), words)
    out = torch.empty((pos.numel(), words), dtype=torch.float32, device=v.device)
    L.check(L.load().gnnmp_gather_f32(L.ptr(src), L.ptr(pos), _ib(pos), index_base, pos.numel(), L.ptr(out), words,
                                      L.stream_ptr()))
    return out.view(v.dtype).view(pos.numel())


def _out_plan(g: GNNGraph) -> Plan:
    key = ("T", False)
    p = g._plans.get(key)
    if p is None:
        p = Plan(g.t, g.s, g.num_nodes, g.num_nodes, g.index_base, False, validate=False)
        g._plans[key] = p
    return p


def sample_neighbors(g: GNNGraph, nodes, K: int = -1, dir: str = "in", replace: bool = False, dropnodes: bool = False,
                     seed: int = 0):
    """sample_neighbors(g, nodes, K; dir, replace) — sampling.jl:68-119 with dropnodes = false: a graph on the same
    node set holding, for every seed node, K of its incoming (dir = "in") or outgoing ("out") edges drawn uniformly
    (all of them if K <= 0 or, without replacement, if it has fewer).  `.eid` of the result holds the positions of the
    kept edges in g (the reference's edata.EID).  The draw is reproducible in `seed`; it is not Julia's RNG stream."""
    assert dir in ("in", "out")
    if dropnodes:
        raise NotImplementedError("sample_neighbors(dropnodes = true) relabels nodes on the host in the reference; "
                                  "not on the device path yet")
    nodes = nodes.to(device=g.device, dtype=g.s.dtype).contiguous()
    plan = g.plan(False) if dir == "in" else _out_plan(g)
    M = nodes.numel()
    lib = L.load()
    offsets = torch.empty(M + 1, dtype=torch.int64, device=g.device)
    cap = M * K if K > 0 else g.num_edges
    eids = torch.empty(max(cap, 1), dtype=g.s.dtype, device=g.device)
    total = ctypes.c_int64(0)
    L.check(lib.gnnmp_sample_neighbors(plan.handle, L.ptr(nodes), g.idx_bytes, g.index_base, M, int(K), int(bool(replace)),
                                       ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), L.ptr(offsets), L.ptr(eids), cap,
                                       ctypes.byref(total), L.stream_ptr()))
    eids = eids[: total.value]
    s = _take_index(g.s, eids, g.index_base)
    t = _take_index(g.t, eids, g.index_base)
    w = None
    if g.w is not None:
        w = torch.empty(eids.numel(), dtype=torch.float32, device=g.device)
        L.check(lib.gnnmp_gather_f32(L.ptr(g.w), L.ptr(eids), g.idx_bytes, g.index_base, eids.numel(), L.ptr(w), 1,
                                     L.stream_ptr()))
    gnew = GNNGraph(s, t, w, num_nodes=g.num_nodes, graph_indicator=g.graph_indicator, num_graphs=g.num_graphs, x=g.x,
                    index_base=g.index_base, device=g.device, _validated=True)
    gnew.eid = eids
    gnew.sample_offsets = offsets
    return gnew
