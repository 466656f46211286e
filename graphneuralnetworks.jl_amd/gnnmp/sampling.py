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
    if pos.numel() == 0 or v.numel() == 0:
        return torch.empty(0, dtype=v.dtype, device=v.device)
    words = 2 if v.dtype == torch.int64 else 1
    src = v.contiguous().view(torch.float32).view(v.numel(), words)
    out = torch.empty((pos.numel(), words), dtype=torch.float32, device=v.device)
    L.check(L.load().gnnmp_gather_f32(L.ptr(src), L.ptr(pos), _ib(pos), index_base, pos.numel(), L.ptr(out), words,
                                      L.stream_ptr()))
    return out.view(v.dtype).view(pos.numel())


def _out_plan(g: GNNGraph) -> Plan:
    return g.plan_transposed(False)


def sample_neighbors(g: GNNGraph, nodes, K: int = -1, dir: str = "in", replace: bool = False, dropnodes: bool = False,
                     seed: int = 0):
    """sample_neighbors(g, nodes, K; dir, replace, dropnodes) — sampling.jl:68-119: a graph holding, for every seed
    node, K of its incoming (dir = "in") or outgoing ("out") edges drawn uniformly (all of them if K <= 0 or, without
    replacement, if it has fewer) — on the same node set, or with dropnodes = true on the seeds followed by the sampled
    neighbours in first-appearance order (`.nid` = the reference's ndata.NID).  `.eid` of the result holds the
    positions of the kept edges in g (edata.EID).  The draw is reproducible in `seed`; it is not Julia's RNG stream."""
    assert dir in ("in", "out")
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
    if not dropnodes:
        gnew = GNNGraph(s, t, w, num_nodes=g.num_nodes, graph_indicator=g.graph_indicator, num_graphs=g.num_graphs, x=g.x,
                        index_base=g.index_base, device=g.device, _validated=True)
    else:
        # sampling.jl:100-116: nodes_all = [nodes; setdiff(s | t, nodes)] (first-appearance order), edges relabelled by
        # position in nodes_all, node data restricted to nodes_all, ndata.NID = nodes_all.  The ordered device set gives
        # exactly that list and its inverse map (1-based positions, 0 = absent).
        ns = NodeSet(g)
        ns.add(nodes)
        assert ns.nodes.numel() == nodes.numel(), "sample_neighbors(dropnodes = true): `nodes` must not repeat a node"
        ns.add(s if dir == "in" else t)
        relabel = lambda v: _take_index(ns.map, v, g.index_base).to(g.s.dtype) - (1 - g.index_base)   # index plumbing
        from .msgpass import _gather
        gi = None if g.graph_indicator is None else _take_index(g.graph_indicator, ns.nodes, g.index_base)
        x = None if g.x is None else _gather(g.x, ns.nodes, g.index_base)
        gnew = GNNGraph(relabel(s), relabel(t), w, num_nodes=ns.nodes.numel(), graph_indicator=gi, num_graphs=g.num_graphs,
                        x=x, index_base=g.index_base, device=g.device, _validated=True)
        gnew.nid = ns.nodes
    gnew.eid = eids
    gnew.sample_offsets = offsets
    return gnew


class NodeSet:
    """An ordered set of nodes on the device: `nodes` (list, discovery order) + map[N] (0 = absent, else 1-based position)."""

    def __init__(self, g: GNNGraph):
        self.g = g
        self.map = torch.zeros(g.num_nodes, dtype=torch.int32, device=g.device)
        self._first = torch.empty(g.num_nodes, dtype=torch.int32, device=g.device)
        self.nodes = torch.empty(0, dtype=g.s.dtype, device=g.device)

    def add(self, cand):
        """append the not-yet-present nodes of `cand` (each once, first occurrence first); returns them"""
        g = self.g
        cand = cand.to(device=g.device, dtype=g.s.dtype).contiguous()
        if cand.numel() == 0:
            return cand
        new = torch.empty(cand.numel(), dtype=g.s.dtype, device=g.device)
        n_new = ctypes.c_int64(0)
        L.check(L.load().gnnmp_unique_append(L.ptr(self.map), L.ptr(self._first), g.num_nodes, L.ptr(cand), g.idx_bytes,
                                             g.index_base, cand.numel(), self.nodes.numel(), L.ptr(new),
                                             ctypes.byref(n_new), L.stream_ptr()))
        new = new[: n_new.value]
        self.nodes = torch.cat([self.nodes, new])
        return new


def induced_subgraph(g: GNNGraph, nodes):
    """Graphs.induced_subgraph(g, nodes) — sampling.jl:173-203: the listed nodes (relabelled by list position) and every
    edge of g between two of them, grouped by target in list order, in-edge order inside a target.  `.nid` / `.eid` of
    the result are the reference's node list / edata indices; node features follow the nodes."""
    ns = nodes if isinstance(nodes, NodeSet) else None
    if ns is None:
        ns = NodeSet(g)
        given = nodes.to(device=g.device, dtype=g.s.dtype).contiguous()
        ns.add(given)
        assert ns.nodes.numel() == given.numel(), "induced_subgraph: the node list must not repeat a node"
    M = ns.nodes.numel()
    lib = L.load()
    offsets = torch.empty(M + 1, dtype=torch.int64, device=g.device)
    total = ctypes.c_int64(0)
    plan = g.plan(False)
    # count-only call first (capacity 0, no outputs), then exactly-sized outputs
    L.check(lib.gnnmp_induced_subgraph(plan.handle, L.ptr(ns.map), L.ptr(ns.nodes), g.idx_bytes, g.index_base, M,
                                       L.ptr(offsets), None, None, None, 0, ctypes.byref(total), L.stream_ptr()))
    E = total.value
    s = torch.empty(E, dtype=g.s.dtype, device=g.device)
    t = torch.empty_like(s)
    eid = torch.empty_like(s)
    if E > 0:
        L.check(lib.gnnmp_induced_subgraph(plan.handle, L.ptr(ns.map), L.ptr(ns.nodes), g.idx_bytes, g.index_base, M,
                                           L.ptr(offsets), L.ptr(s), L.ptr(t), L.ptr(eid), E, ctypes.byref(total),
                                           L.stream_ptr()))
    x = None
    if g.x is not None:
        from .msgpass import _gather
        x = _gather(g.x, ns.nodes, g.index_base)
    w = None
    if g.w is not None and E > 0:
        w = torch.empty(E, dtype=torch.float32, device=g.device)
        L.check(lib.gnnmp_gather_f32(L.ptr(g.w), L.ptr(eid), g.idx_bytes, g.index_base, E, L.ptr(w), 1, L.stream_ptr()))
    sub = GNNGraph(s, t, w, num_nodes=M, x=x, index_base=g.index_base, device=g.device, _validated=True)
    sub.nid, sub.eid = ns.nodes, eid
    return sub


class NeighborLoader:
    """NeighborLoader(graph; num_neighbors, input_nodes, num_layers, batch_size) — GNNGraphs/src/samplers.jl:27-101: for
    every batch of input nodes, `num_layers` rounds of "sample num_neighbors[l] in-neighbours of every frontier node (with
    replacement, as `rand(neighbors, k)` does), the sampled nodes become the next frontier", then the induced subgraph of
    everything reached.  All steps run on the device (sample -> set union -> induced subgraph).  Deviation, documented:
    the reference expands each input node's neighbourhood separately (a node reached from two input nodes is expanded
    twice, independently); here the batch shares one frontier per layer.  Node order inside a mini-batch is discovery
    order (the reference's is the iteration order of a Julia `Set`, i.e. unspecified)."""

    def __init__(self, graph: GNNGraph, num_neighbors, num_layers, input_nodes=None, batch_size=None, seed=0):
        assert len(num_neighbors) >= num_layers
        self.graph, self.num_neighbors, self.num_layers, self.seed = graph, list(num_neighbors), int(num_layers), int(seed)
        if input_nodes is None:
            input_nodes = torch.arange(graph.index_base, graph.num_nodes + graph.index_base, dtype=graph.s.dtype,
                                       device=graph.device)
        self.input_nodes = input_nodes.to(device=graph.device, dtype=graph.s.dtype).contiguous()
        self.batch_size = int(batch_size) if batch_size is not None else max(1, self.input_nodes.numel())

    def __len__(self):
        return (self.input_nodes.numel() + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        g = self.graph
        lib = L.load()
        for b, start in enumerate(range(0, self.input_nodes.numel(), self.batch_size)):
            batch = self.input_nodes[start:start + self.batch_size]
            sub_nodes = NodeSet(g)
            frontier = sub_nodes.add(batch)
            frontier = sub_nodes.nodes                      # a repeated input node is one node
            for layer in range(self.num_layers):
                K = self.num_neighbors[layer]
                if K <= 0 or frontier.numel() == 0:
                    frontier = frontier[:0]
                    continue
                # rand(neighbors, min(K, d)): d picks at most, drawn with replacement
                plan = g.plan(False)
                M = frontier.numel()
                offsets = torch.empty(M + 1, dtype=torch.int64, device=g.device)
                eids = torch.empty(M * K, dtype=g.s.dtype, device=g.device)
                total = ctypes.c_int64(0)
                L.check(lib.gnnmp_sample_neighbors(plan.handle, L.ptr(frontier), g.idx_bytes, g.index_base, M, K, 2,
                                                   ctypes.c_uint64((self.seed * 1000003 + b * 101 + layer) & (2**64 - 1)),
                                                   L.ptr(offsets), L.ptr(eids), M * K, ctypes.byref(total), L.stream_ptr()))
                cand = _take_index(g.s, eids[: total.value], g.index_base)
                layer_set = NodeSet(g)                      # the sampled nodes, each once: the next frontier
                frontier = layer_set.add(cand)
                sub_nodes.add(frontier)
            yield induced_subgraph(g, sub_nodes)
