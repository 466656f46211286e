"""GNNGraph{COO_T} and the graph queries / transforms that sit on the hot path.

Mirror of the reference container and functions (same names, argument meaning and error behaviour):
  GNNGraph            GNNGraphs/src/gnngraph.jl:108-117 (COO storage `graph = (s, t, w | nothing)`)
  edge_index          GNNGraphs/src/query.jl:12
  degree              GNNGraphs/src/query.jl:314-331, 355-369
  graph_indicator     GNNGraphs/src/query.jl:500-512
  add_self_loops      GNNGraphs/src/transform.jl:12-28
  set_edge_weight     GNNGraphs/src/transform.jl:568-577
  batch               GNNGraphs/src/transform.jl:682-709 (MLUtils.batch)
  check_num_nodes / check_num_edges   GNNGraphs/src/utils.jl:1-28  (AssertionError, like Julia's @assert)

Feature arrays are torch float32 tensors shaped [N, ...] (= Julia (..., N) column-major).  Index vectors are kept
exactly as the reference holds them: 1-based Int64 (or Int32) device vectors; `index_base=0` is accepted for
Python callers.  All arithmetic happens in libgnnmp (HIP); torch only owns the memory.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L


def _as_index(v, device):
    if isinstance(v, torch.Tensor):
        t = v
    else:
        t = torch.as_tensor(v)
    if t.dtype not in (torch.int64, torch.int32):
        t = t.to(torch.int64)
    return t.to(device).contiguous()


def _as_f32(v, device):
    if v is None:
        return None
    t = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
    return t.to(device=device, dtype=torch.float32).contiguous()


class Plan:
    """Owner of one gnnmp_graph_t (dst-sorted CSR of an edge index); freed with the Python object."""

    def __init__(self, s, t, n_src, n_dst, index_base, add_self_loops, validate=True):
        L.require_gpu()
        lib = L.load()
        self._h = ctypes.c_void_p()
        self._lib = lib
        idx_bytes = 8 if s.dtype == torch.int64 else 4
        rc = lib.gnnmp_plan_create(ctypes.byref(self._h), L.ptr(s), L.ptr(t), idx_bytes, index_base,
                                   n_src, n_dst, s.numel(), 1 if add_self_loops else 0,
                                   1 if validate else 0, L.stream_ptr())
        if rc == L.EBOUNDS:
            # GNNGraphs/src/convert.jl:47-54 asserts the index range at construction
            raise AssertionError(lib.gnnmp_last_error().decode())
        L.check(rc)
        info = (ctypes.c_int64 * 8)()
        L.check(lib.gnnmp_plan_info(self._h, info))
        self.n_src, self.n_dst, self.n_edges, self.n_total = info[0], info[1], info[2], info[3]
        self.max_degree, self.n_long, self.bytes, self.long_thresh = info[4], info[5], info[6], info[7]
        self.device = s.device

    @classmethod
    def from_csc(cls, colptr, rowval, n_src, n_dst, index_base=1, validate=True):
        """the plan of a sparse-matrix graph (gnnmp_plan_from_csc): the CSC structure IS the dst-sorted CSR, no sort"""
        L.require_gpu()
        self = object.__new__(cls)
        self._lib = L.load()
        self._h = ctypes.c_void_p()
        idx_bytes = 8 if colptr.dtype == torch.int64 else 4
        rc = self._lib.gnnmp_plan_from_csc(ctypes.byref(self._h), L.ptr(colptr), L.ptr(rowval), idx_bytes, index_base, n_src, n_dst,
                                           rowval.numel(), 1 if validate else 0, L.stream_ptr())
        if rc == L.EBOUNDS:
            raise AssertionError(self._lib.gnnmp_last_error().decode())
        L.check(rc)
        info = (ctypes.c_int64 * 8)()
        L.check(self._lib.gnnmp_plan_info(self._h, info))
        self.n_src, self.n_dst, self.n_edges, self.n_total = info[0], info[1], info[2], info[3]
        self.max_degree, self.n_long, self.bytes, self.long_thresh = info[4], info[5], info[6], info[7]
        self.device = colptr.device
        return self

    @classmethod
    def _adopt(cls, handle, device):
        """wrap a handle made by gnnmp_plan_concat / gnnmp_plan_select (a POOLED plan: released stream-ordered, no host sync)"""
        self = object.__new__(cls)
        self._lib = L.load()
        self._h = handle
        self._pooled = True
        self._last_stream = torch.cuda.current_stream()      # the stream gnnmp_plan_concat / _select ran on
        info = (ctypes.c_int64 * 8)()
        L.check(self._lib.gnnmp_plan_info(self._h, info))
        self.n_src, self.n_dst, self.n_edges, self.n_total = info[0], info[1], info[2], info[3]
        self.max_degree, self.n_long, self.bytes, self.long_thresh = info[4], info[5], info[6], info[7]
        self.device = device
        return self

    @property
    def handle(self):
        # every compute call reads the handle right before it passes the current stream to the library: remember that stream, so that a
        # pooled plan is released BEHIND its last use even when the garbage collector runs while another stream is current
        if getattr(self, "_pooled", False):
            self._last_stream = torch.cuda.current_stream()
        return self._h

    def status(self):
        """pooled plans: raises if the member table did not match the announced totals (synchronises the stream)"""
        L.check(self._lib.gnnmp_plan_status(self._h, L.stream_ptr()))

    def edge_index(self, dtype=torch.int64, index_base=1):
        """(s, t) of the plan's graph in original edge order (gnnmp_plan_edge_index)"""
        s = torch.empty(self.n_edges, dtype=dtype, device=self.device)
        t = torch.empty(self.n_edges, dtype=dtype, device=self.device)
        L.check(self._lib.gnnmp_plan_edge_index(self._h, 8 if dtype == torch.int64 else 4, index_base, L.ptr(s), L.ptr(t), L.stream_ptr()))
        return s, t

    def export(self):
        """(rowptr, col, eid) int32 device tensors — the plan's bit-exact index outputs (plans of fewer than 2^31 slots)"""
        rowptr = torch.empty(self.n_dst + 1, dtype=torch.int32, device=self.device)
        col = torch.empty(self.n_total, dtype=torch.int32, device=self.device)
        eid = torch.empty(self.n_total, dtype=torch.int32, device=self.device)
        L.check(self._lib.gnnmp_plan_export(self._h, L.ptr(rowptr), L.ptr(col), L.ptr(eid), L.stream_ptr()))
        return rowptr, col, eid

    def export64(self):
        """(rowptr int64, col int32, eid as int64) — the arrays as stored, at any plan size; eid is held as unsigned 32-bit on the
        device and widened here (torch has no uint32 arithmetic)"""
        rowptr = torch.empty(self.n_dst + 1, dtype=torch.int64, device=self.device)
        col = torch.empty(self.n_total, dtype=torch.int32, device=self.device)
        eid = torch.empty(self.n_total, dtype=torch.int32, device=self.device)
        L.check(self._lib.gnnmp_plan_export64(self._h, L.ptr(rowptr), L.ptr(col), L.ptr(eid), L.stream_ptr()))
        return rowptr, col, eid.to(torch.int64) & 0xFFFFFFFF

    def __del__(self):
        try:
            if self._h:
                if getattr(self, "_pooled", False):
                    # stream-ordered: the block goes back to the library's pool behind the work enqueued so far ON THE STREAM THE PLAN
                    # WAS LAST USED ON (no host synchronisation); a plan that never saw a stream of ours takes the unknown-streams path
                    st = getattr(self, "_last_stream", None)
                    if st is not None:
                        self._lib.gnnmp_plan_release(self._h, ctypes.c_void_p(st.cuda_stream))
                    else:
                        self._lib.gnnmp_plan_destroy(self._h)
                else:
                    self._lib.gnnmp_plan_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


class GNNGraph:
    """COO graph: `GNNGraph(s, t)` / `GNNGraph((s, t, w))` — GNNGraphs/src/gnngraph.jl:108-117.

    s, t       : source / target node of each edge (1-based by default, like Julia)
    w          : optional edge weights (Float32[E])
    num_nodes  : defaults to max(maximum(s), maximum(t))  (convert.jl:33-36)
    ndata `x`  : optional node features [num_nodes, ...]
    """

    def __init__(self, s, t=None, w=None, num_nodes=None, graph_indicator=None, num_graphs=1, x=None,
                 index_base=1, device=None, _validated=False):
        L.require_gpu()
        if t is None and isinstance(s, (tuple, list)) and len(s) in (2, 3) and not isinstance(s[0], (int, float)):
            tup = s
            s, t = tup[0], tup[1]
            w = tup[2] if len(tup) == 3 else w
        device = torch.device(device) if device is not None else (
            s.device if isinstance(s, torch.Tensor) and s.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        self.index_base = int(index_base)
        assert self.index_base in (0, 1)
        s = _as_index(s, device)
        t = _as_index(t, device)
        if s.dtype != t.dtype:
            t = t.to(s.dtype)
        assert s.dim() == 1 and t.dim() == 1 and s.numel() == t.numel(), "length(s) == length(t)"
        self.w = _as_f32(w, device)
        assert self.w is None or self.w.numel() == s.numel(), "length(val) == length(s)"
        if num_nodes is None:
            num_nodes = 0 if s.numel() == 0 else int(max(int(s.max()), int(t.max()))) + (1 - self.index_base)
        self._s, self._t = s, t
        self.num_nodes = int(num_nodes)
        self.num_edges = int(s.numel())
        self.num_graphs = int(num_graphs)
        self.graph_indicator = None if graph_indicator is None else _as_index(graph_indicator, device)
        self.x = _as_f32(x, device)
        if self.x is not None:
            check_num_nodes(self, self.x)
        self.device = device
        self._plans = {}
        self._cache = {}   # per-graph constants in plan slot order (GCN normalisation, graph weights)
        # True once a validating plan build (or the caller, for indices the library itself produced) has established
        # base <= s, t < n + base (convert.jl:47-54).  Every plan built while it is False validates.
        self._indices_validated = bool(_validated)
        if not _validated:
            self.plan(False)  # builds the CSR plan and validates 1 <= s,t <= n (convert.jl:47-54)

    @classmethod
    def _from_plan(cls, plan: "Plan", num_graphs, graph_indicator, x, index_base, idx_dtype, self_loops=False):
        """A graph that exists as its PLAN (gnnmp_plan_select / gnnmp_plan_concat: the batch of a training step): s and t are
        materialised from the plan only if somebody asks for them (gnnmp_plan_edge_index)."""
        g = object.__new__(cls)
        g.index_base = int(index_base)
        g._s = g._t = None
        g._idx_dtype = idx_dtype
        g.w = None
        g.num_nodes = int(plan.n_dst)
        g.num_edges = int(plan.n_edges)
        g.num_graphs = int(num_graphs)
        g.graph_indicator = graph_indicator
        g.x = x
        g.device = plan.device
        g._plans = {bool(self_loops): plan}
        g._cache = {}
        g._indices_validated = True
        return g

    @classmethod
    def from_sparse(cls, A=None, colptr=None, rowval=None, nzval=None, num_nodes=None, index_base=1, x=None, device=None):
        """`GNNGraph(A::AbstractSparseMatrix)` — a graph of type :sparse (GNNGraphs/src/gnngraph.jl:108, abstracttypes.jl:5):
        A[s, t] != 0 is an edge s -> t, and edge_index(g) = findnz(A) walks A's columns (query.jl:14, convert.jl:62-73), so the edges
        are destination-sorted as stored and the plan is A's CSC structure itself (gnnmp_plan_from_csc: no sort).
        get_edge_weight(g) = nzval (query.jl:18).  `A`: a scipy.sparse matrix or a torch sparse CSC / CSR / COO tensor; or the three
        CSC arrays (colptr / rowval in `index_base`)."""
        L.require_gpu()
        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if A is not None:
            if isinstance(A, torch.Tensor):
                A = A.to_sparse_csc() if A.layout != torch.sparse_csc else A
                n = int(A.shape[0])
                assert A.shape[0] == A.shape[1], "adjacency matrix must be square"
                colptr, rowval, nzval = A.ccol_indices() + index_base, A.row_indices() + index_base, A.values()
            else:                                   # scipy.sparse
                A = A.tocsc()
                A.sort_indices()
                n = int(A.shape[0])
                assert A.shape[0] == A.shape[1], "adjacency matrix must be square"
                import numpy as _np
                colptr = torch.from_numpy(A.indptr.astype(_np.int64) + index_base)
                rowval = torch.from_numpy(A.indices.astype(_np.int64) + index_base)
                nzval = torch.from_numpy(A.data.astype(_np.float32))
            num_nodes = n if num_nodes is None else num_nodes
        colptr, rowval = _as_index(colptr, device), _as_index(rowval, device)
        if rowval.dtype != colptr.dtype:
            rowval = rowval.to(colptr.dtype)
        if num_nodes is None:
            num_nodes = int(colptr.numel()) - 1
        assert colptr.numel() == num_nodes + 1, "length(colptr) == num_nodes + 1"
        plan = Plan.from_csc(colptr, rowval, num_nodes, num_nodes, index_base)
        g = cls._from_plan(plan, 1, None, None, index_base, colptr.dtype)
        g.graph_type = "sparse"
        g.w = _as_f32(nzval, device)
        assert g.w is None or g.w.numel() == g.num_edges, "length(nzval) == nnz"
        g.x = _as_f32(x, device)
        if g.x is not None:
            check_num_nodes(g, g.x)
        return g

    def _materialise(self):
        p = self._plans.get(False) or self._plans.get(True)
        self._s, self._t = p.edge_index(self._idx_dtype, self.index_base)

    @property
    def s(self):
        if self._s is None:
            self._materialise()
        return self._s

    @s.setter
    def s(self, v):
        self._s = v

    @property
    def t(self):
        if self._t is None:
            self._materialise()
        return self._t

    @t.setter
    def t(self, v):
        self._t = v

    # -- plans (cached; the reference rebuilds sparse(s,t,...) on every call, convert.jl:221-237) --
    def plan(self, add_self_loops: bool = False) -> Plan:
        key = bool(add_self_loops)
        p = self._plans.get(key)
        if p is None:
            p = Plan(self.s, self.t, self.num_nodes, self.num_nodes, self.index_base, key,
                     validate=not self._indices_validated)
            self._indices_validated = True
            self._plans[key] = p
        return p

    def plan_transposed(self, add_self_loops: bool = False) -> Plan:
        """plan of the reversed edge index (t, s): row j lists the edges that LEAVE j, in original edge order (out-degrees,
        the adjoints' walk, out-neighbour sampling)"""
        key = ("T", bool(add_self_loops))
        p = self._plans.get(key)
        if p is None:
            p = Plan(self.t, self.s, self.num_nodes, self.num_nodes, self.index_base, key[1],
                     validate=not self._indices_validated)
            self._indices_validated = True
            self._plans[key] = p
        return p

    @property
    def idx_bytes(self):
        dt = self._s.dtype if self._s is not None else self._idx_dtype
        return 8 if dt == torch.int64 else 4

    graph_type = "coo"          # get_graph_type (GNNGraphs/src/query.jl:97-99); "sparse" for from_sparse graphs

    def __repr__(self):
        return f"GNNGraph(num_nodes={self.num_nodes}, num_edges={self.num_edges}, num_graphs={self.num_graphs})"


# ---------------------------------------------------------------------------------------------------------
# checks — GNNGraphs/src/utils.jl:1-28
# ---------------------------------------------------------------------------------------------------------
def check_num_nodes(g: GNNGraph, x):
    if x is None:
        return True
    if isinstance(x, dict):
        for v in x.values():
            check_num_nodes(g, v)
        return True
    if isinstance(x, (tuple, list)):
        for v in x:
            check_num_nodes(g, v)
        return True
    assert g.num_nodes == x.shape[0], \
        f"Got {x.shape[0]} as last dimension size instead of num_nodes={g.num_nodes}"
    return True


def check_num_edges(g: GNNGraph, e):
    if e is None:
        return True
    if isinstance(e, dict):
        for v in e.values():
            check_num_edges(g, v)
        return True
    if isinstance(e, (tuple, list)):
        for v in e:
            check_num_edges(g, v)
        return True
    assert g.num_edges == e.shape[0], \
        f"Got {e.shape[0]} as last dimension size instead of num_edges={g.num_edges}"
    return True


# ---------------------------------------------------------------------------------------------------------
# queries
# ---------------------------------------------------------------------------------------------------------
def edge_index(g: GNNGraph):
    """(s, t) — zero-copy for COO graphs (GNNGraphs/src/query.jl:12)"""
    return g.s, g.t


def get_edge_weight(g: GNNGraph):
    return g.w


def get_graph_type(g: GNNGraph):
    """:coo / :sparse — GNNGraphs/src/query.jl:97-99"""
    return g.graph_type


def graph_indicator(g: GNNGraph, edges: bool = False):
    """GNNGraphs/src/query.jl:500-512"""
    gi = g.graph_indicator
    if gi is None:
        gi = torch.full((g.num_nodes,), g.index_base, dtype=g.s.dtype, device=g.device)
    if edges:
        # graph_indicator(g)[s] (query.jl:507-509): a bit copy of index words, done by the float gather on a reinterpreted
        # view (one or two 32-bit words per index)
        ge = g._cache.get("edge_indicator")
        if ge is None:
            words = 2 if gi.dtype == torch.int64 else 1
            src = gi.contiguous().view(torch.float32).view(g.num_nodes, words)
            out = torch.empty((g.num_edges, words), dtype=torch.float32, device=g.device)
            L.check(L.load().gnnmp_gather_f32(L.ptr(src), L.ptr(g.s), g.idx_bytes, g.index_base, g.num_edges, L.ptr(out),
                                              words, L.stream_ptr()))
            ge = out.view(gi.dtype).view(g.num_edges)
            g._cache["edge_indicator"] = ge
        return ge
    return gi


def degree(g: GNNGraph, T=None, dir: str = "out", edge_weight=True):
    """degree(g, T = nothing; dir = :out, edge_weight = true) — GNNGraphs/src/query.jl:314-331,355-369: same defaults.

    edge_weight: True (use the graph's weights if any), False/None (count edges) or a weight vector.
    T: None follows the reference's typing (query.jl:336-345): the weights' element type (Float32) when weights are used,
    the index element type (Int64 / Int32) when edges are counted.  The count itself runs in Float32 on the device (the
    element type gcn_conv asks for, conv.jl:43,53-55) and is exact below 2^24 edges per node."""
    assert dir in ("in", "out", "both")
    if isinstance(edge_weight, torch.Tensor) or isinstance(edge_weight, (list, tuple)):
        w = _as_f32(edge_weight, g.device)
        assert w.numel() == g.num_edges
    elif edge_weight is True:
        w = g.w
    else:
        w = None
    out = None
    lib = L.load()
    for d in (("in", "out") if dir == "both" else (dir,)):
        if d == "in":
            plan = g.plan(False)
        else:
            plan = g.plan_transposed()
        deg = torch.empty(g.num_nodes, dtype=torch.float32, device=g.device)
        L.check(lib.gnnmp_degree_f32(plan.handle, L.ptr(w), L.ptr(deg), L.stream_ptr()))
        if out is None:
            out = deg
        else:
            both = torch.empty_like(deg)
            L.check(lib.gnnmp_add_f32(L.ptr(out), L.ptr(deg), L.ptr(both), deg.numel(), L.stream_ptr()))
            out = both
    if T is None:
        T = torch.float32 if w is not None else g.s.dtype
    if T != torch.float32:
        out = out.to(T)
    return out


# ---------------------------------------------------------------------------------------------------------
# transforms
# ---------------------------------------------------------------------------------------------------------
def add_self_loops(g: GNNGraph) -> GNNGraph:
    """add_self_loops(g::GNNGraph{COO}) — GNNGraphs/src/transform.jl:12-28: appends (i, i) for every node, after
    the existing edges, never de-duplicating; appended weights are 1."""
    n, E = g.num_nodes, g.num_edges
    s2 = torch.empty(E + n, dtype=g.s.dtype, device=g.device)
    t2 = torch.empty(E + n, dtype=g.s.dtype, device=g.device)
    w2 = None if g.w is None else torch.empty(E + n, dtype=torch.float32, device=g.device)
    L.check(L.load().gnnmp_add_self_loops(L.ptr(g.s), L.ptr(g.t), g.idx_bytes, g.index_base, E, n, L.ptr(s2),
                                          L.ptr(t2), L.ptr(g.w), L.ptr(w2), L.stream_ptr()))
    return GNNGraph(s2, t2, w2, num_nodes=n, graph_indicator=g.graph_indicator, num_graphs=g.num_graphs, x=g.x,
                    index_base=g.index_base, device=g.device, _validated=True)


def set_edge_weight(g: GNNGraph, w) -> GNNGraph:
    """GNNGraphs/src/transform.jl:568-577"""
    w = _as_f32(w, g.device)
    assert w.numel() == g.num_edges
    g2 = GNNGraph(g.s, g.t, w, num_nodes=g.num_nodes, graph_indicator=g.graph_indicator,
                  num_graphs=g.num_graphs, x=g.x, index_base=g.index_base, device=g.device, _validated=True)
    g2._plans = g._plans  # same (s, t): plans are shared, like the reference shares s, t across copies
    return g2


def batch_arrays(members, xs=None, index_base=1, device=None) -> GNNGraph:
    """MLUtils.batch for member graphs given as host records (s, t, num_nodes) with LOCAL numbering (what a DataLoader
    collates, GNNGraphs/src/transform.jl:682-709): one upload of the concatenated indices, offsets and graph_indicator
    computed on the device by gnnmp_batch_coo.  xs: optional list of per-graph feature arrays."""
    import numpy as np
    L.require_gpu()
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    G = len(members)
    if G == 0:
        raise ValueError("Cannot batch an empty vector of graphs")
    ne = np.zeros(G + 1, np.int64)
    nn = np.zeros(G + 1, np.int64)
    ne[1:] = np.cumsum([len(m[0]) for m in members])
    nn[1:] = np.cumsum([int(m[2]) for m in members])
    s_host = np.concatenate([np.asarray(m[0], np.int64) for m in members])
    t_host = np.concatenate([np.asarray(m[1], np.int64) for m in members])
    # every member's indices against ITS OWN num_nodes, as the reference asserts when each member graph is constructed
    # (convert.jl:47-54): after the offsets are added an out-of-range index could land inside a neighbour's node range
    hi = np.repeat(np.diff(nn), np.diff(ne)) + (index_base - 1)
    for name, v in (("s", s_host), ("t", t_host)):
        bad = np.nonzero((v < index_base) | (v > hi))[0]
        if bad.size:
            k = int(bad[0])
            gi_bad = int(np.searchsorted(ne, k, side="right") - 1)
            raise AssertionError(f"batch: {name}[{k - int(ne[gi_bad])}] = {int(v[k])} of member graph {gi_bad} is outside "
                                 f"{index_base}..{int(hi[k])}")
    s_cat = torch.from_numpy(s_host).to(device)
    t_cat = torch.from_numpy(t_host).to(device)
    ned, nnd = torch.from_numpy(ne).to(device), torch.from_numpy(nn).to(device)
    s2, t2 = torch.empty_like(s_cat), torch.empty_like(t_cat)
    gi = torch.empty(int(nn[-1]), dtype=torch.int64, device=device)
    L.check(L.load().gnnmp_batch_coo(L.ptr(s_cat), L.ptr(t_cat), 8, index_base, L.ptr(ned), L.ptr(nnd), G, L.ptr(s2),
                                     L.ptr(t2), L.ptr(gi), L.stream_ptr()))
    x = None
    if xs is not None:
        x = torch.from_numpy(np.concatenate([np.asarray(v, np.float32) for v in xs])).to(device)
    return GNNGraph(s2, t2, None, num_nodes=int(nn[-1]), graph_indicator=gi, num_graphs=G, x=x, index_base=index_base,
                    device=device, _validated=True)


def batch(gs) -> GNNGraph:
    """MLUtils.batch(::Vector{GNNGraph{COO}}) — GNNGraphs/src/transform.jl:682-709: concatenates the edge indices with
    node offsets cumsum(num_nodes), builds graph_indicator, concatenates node features."""
    gs = list(gs)
    if len(gs) == 0:
        raise ValueError("Cannot batch an empty vector of graphs")
    g0 = gs[0]
    dev, base, dt = g0.device, g0.index_base, g0.s.dtype
    for g in gs:
        if not isinstance(g, GNNGraph):
            raise ValueError("Cannot batch a non-GNNGraph")  # ArgumentError, transform.jl:711-713
        assert g.index_base == base and g.s.dtype == dt
        assert g.num_graphs == 1 or g.graph_indicator is not None
    if any(g.num_graphs != 1 for g in gs):
        # nested batches: flatten through their own indicators (graphsum offsets, transform.jl:697-699)
        raise NotImplementedError("batching already-batched graphs is outside the hot path")
    import itertools
    ne = torch.tensor([0] + list(itertools.accumulate(g.num_edges for g in gs)), dtype=torch.int64, device=dev)
    nn = torch.tensor([0] + list(itertools.accumulate(g.num_nodes for g in gs)), dtype=torch.int64, device=dev)
    Etot, Ntot = int(ne[-1]), int(nn[-1])
    s_cat = torch.cat([g.s for g in gs])
    t_cat = torch.cat([g.t for g in gs])
    s2, t2 = torch.empty_like(s_cat), torch.empty_like(t_cat)
    gi = torch.empty(Ntot, dtype=dt, device=dev)
    L.check(L.load().gnnmp_batch_coo(L.ptr(s_cat), L.ptr(t_cat), g0.idx_bytes, base, L.ptr(ne), L.ptr(nn), len(gs),
                                     L.ptr(s2), L.ptr(t2), L.ptr(gi), L.stream_ptr()))
    ws = [g.w for g in gs]
    w = torch.cat(ws) if all(x is not None for x in ws) else None
    xs = [g.x for g in gs]
    x = torch.cat(xs) if all(v is not None for v in xs) else None
    out = GNNGraph(s2, t2, w, num_nodes=Ntot, graph_indicator=gi, num_graphs=len(gs), x=x, index_base=base,
                   device=dev, _validated=True)
    assert out.num_edges == Etot
    return out
