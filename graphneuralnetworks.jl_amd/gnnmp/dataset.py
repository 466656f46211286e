"""A dataset of member graphs RESIDENT on the device, and the batches of a training loop taken from it without a sort, a host copy of the
sizes, or a synchronisation.

The reference's graph-classification loop (GraphNeuralNetworks/examples/graph_classification_tudataset.jl:70-71, 97-104) is
    train_loader = DataLoader(train_data; batchsize, shuffle = true, collate = true)
    for (g, y) in train_loader;  g, y = (g, y) |> device;  model(g, g.ndata.x) ...
i.e. MLUtils.batch(gs[idx]) on the CPU (GNNGraphs/src/transform.jl:682-709) and an upload, EVERY step.  With 288 GB of HBM the dataset lives
on the device once — `GraphDataset` = MLUtils.batch(all graphs) + its plan + the node offsets — and a step's batch is
    GraphDataset.batch(ids)  ==  MLUtils.batch(gs[ids])          (members in the order of `ids`; also getobs / getgraph for ascending ids,
                                                                  GNNGraphs/src/gnngraph.jl:311, transform.jl:827-876)
made by gnnmp_plan_select (the batch's dst-sorted CSR is the concatenation of the members' CSRs: two launches), one gnnmp_gather_f32 for
the node features and gnnmp_chain_jobs_pack for the fused chain's wave jobs: no host synchronisation anywhere.  The batch is a GNNGraph
that exists as its plan; `g.s` / `g.t` are materialised only if asked for.  `DataLoader` mirrors MLUtils.DataLoader(data; batchsize,
shuffle, collate = true) over such a dataset: the epoch's permutation is uploaded once, every batch is a device slice of it.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib as L
from .graph import GNNGraph, Plan, batch_arrays


class GraphDataset:
    """MLUtils.batch(all member graphs), resident: `gall` (a batched GNNGraph with graph_indicator and ndata x), host mirrors of every
    member's num_nodes / num_edges (host integers in the reference too, gnngraph.jl:108-117), device node offsets."""

    def __init__(self, gall: GNNGraph, num_nodes=None, num_edges=None, targets=None):
        assert gall.graph_indicator is not None and gall.num_graphs >= 1, "a batched graph (MLUtils.batch) is expected"
        self.gall = gall
        self.device = gall.device
        G = gall.num_graphs
        lib = L.load()
        if num_nodes is None or num_edges is None:
            # one-off: member sizes from the indicator (a sorted indicator is what batch produces)
            gi = gall.graph_indicator.to(torch.int64) - gall.index_base
            num_nodes = torch.bincount(gi, minlength=G).cpu().numpy()
            ge = gi[(gall.t.to(torch.int64) - gall.index_base)]
            num_edges = torch.bincount(ge, minlength=G).cpu().numpy()
        self.nn = np.asarray(num_nodes, np.int64)
        self.ne = np.asarray(num_edges, np.int64)
        assert self.nn.shape == (G,) and self.ne.shape == (G,)
        assert int(self.nn.sum()) == gall.num_nodes and int(self.ne.sum()) == gall.num_edges
        self.node_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(self.nn)])).to(self.device)
        self.plan = gall.plan(False)
        # gnnmp_plan_select's precondition: the dataset's edge list is member-major (what MLUtils.batch produces) — checked once here
        if gall.num_edges > 0:
            gi = gall.graph_indicator
            words = 2 if gi.dtype == torch.int64 else 1
            src = gi.contiguous().view(torch.float32).view(gall.num_nodes, words)
            for idx in (gall.t, gall.s):
                ge = torch.empty((gall.num_edges, words), dtype=torch.float32, device=self.device)
                L.check(lib.gnnmp_gather_f32(L.ptr(src), L.ptr(idx), gall.idx_bytes, gall.index_base, gall.num_edges, L.ptr(ge), words,
                                             L.stream_ptr()))
                ok = ctypes.c_int(0)
                L.check(lib.gnnmp_is_sorted(L.ptr(ge.view(gi.dtype)), 8 if gi.dtype == torch.int64 else 4, gall.num_edges,
                                            ctypes.byref(ok), L.stream_ptr()))
                assert ok.value == 1, "GraphDataset: the edges of the batched graph are not grouped by member graph in member order"
        self.targets = targets          # optional [G, ...] device tensor (the y of (g, y) pairs)
        self.num_graphs = G

    @classmethod
    def from_members(cls, members, xs=None, targets=None, index_base=1, device=None):
        """members: host records (s, t, num_nodes) with local numbering (what a vector of GNNGraphs holds); xs: per-member features"""
        gall = batch_arrays(members, xs, index_base=index_base, device=device)
        nn = np.array([int(m[2]) for m in members], np.int64)
        ne = np.array([len(m[0]) for m in members], np.int64)
        return cls(gall, nn, ne, targets)

    def __len__(self):
        return self.num_graphs

    def batch(self, ids, ids_dev=None, with_x=True) -> GNNGraph:
        """MLUtils.batch(gs[ids]): ids = 0-based member ids on the HOST (numpy / list: the sizes are summed there), ids_dev = the same ids
        on the device as int64 in the graph's index base (ids + index_base; uploaded here when not given)."""
        ids = np.asarray(ids, np.int64)
        k = int(ids.shape[0])
        if k == 0:
            raise ValueError("Cannot batch an empty vector of graphs")
        nn = self.nn[ids]
        n_rows = int(nn.sum())
        n_edges = int(self.ne[ids].sum())
        dev = self.device
        gall = self.gall
        base = gall.index_base
        if ids_dev is None:       # device ids in the graph's own index base (1-based like Julia by default): the indicator comes out in it too
            ids_dev = torch.from_numpy(ids + base).to(self.device)
        seg = torch.empty(k + 1, dtype=torch.int64, device=dev)
        nmap = torch.empty(n_rows, dtype=torch.int32, device=dev)
        gi = torch.empty(n_rows, dtype=torch.int64, device=dev)
        lib = L.load()
        h = ctypes.c_void_p()
        st = L.stream_ptr()
        L.check(lib.gnnmp_plan_select(ctypes.byref(h), self.plan.handle, L.ptr(self.node_ptr), self.num_graphs, L.ptr(ids_dev), 8, base, k,
                                      n_rows, n_edges, L.ptr(seg), L.ptr(nmap), L.ptr(gi), st))
        plan = Plan._adopt(h, dev)
        x = None
        if with_x and gall.x is not None:
            xf = gall.x if gall.x.dim() == 2 else gall.x.reshape(gall.num_nodes, -1)
            D = xf.shape[1]
            x = torch.empty((n_rows, D), dtype=torch.float32, device=dev)
            L.check(lib.gnnmp_gather_f32(L.ptr(xf), L.ptr(nmap), 4, 0, n_rows, L.ptr(x), D, st))
        g = GNNGraph._from_plan(plan, k, gi, x, gall.index_base, gall.s.dtype if gall._s is not None else torch.int64)
        g._cache["node_ptr"] = seg
        g._cache["node_map"] = nmap
        g._cache["member_stats"] = (n_rows, int(nn.max()), bool((nn == 0).any()))     # host-known: lets the chain pack its jobs on the device
        return g

    def targets_of(self, ids_dev):
        """targets of the members `ids_dev` (device ids in the graph's index base)"""
        return None if self.targets is None else self.targets.index_select(0, ids_dev - self.gall.index_base)


class DataLoader:
    """MLUtils.DataLoader(data; batchsize, shuffle, collate = true) over a GraphDataset: yields the batched GNNGraph of every step (and
    the targets when the dataset has them).  The permutation of an epoch is drawn on the host (numpy, seeded) and uploaded ONCE; a batch's
    device ids are a slice of it.

    prefetch = True (round 6): batch k + 1 is PREPARED ON A SIDE STREAM while the consumer's stream runs batch k — its plan
    (gnnmp_plan_select), its features (one gather) and whatever `prepare(g)` adds (gnnmp.layers.chain_prepare: the fused chain's wave jobs)
    are a handful of small latency-bound kernels that fit beside the step's own kernels; the consumer's stream waits for the batch's event
    before it touches it (no host synchronisation), and the first batch of the NEXT epoch is prepared behind the last batch of this one.
    The sequence of batches is the same with and without prefetch (same seeded permutations in the same order).  Blocks of the library's
    pool cross streams by construction (csrc/pool.h: a block is parked with an event on its last stream); torch tensors made under the
    side stream are handed to the consumer's stream with record_stream."""

    def __init__(self, data: GraphDataset, batchsize=1, shuffle=False, partial=True, seed=None, prefetch=False, prepare=None):
        self.data = data
        self.batchsize = int(batchsize)
        self.shuffle = bool(shuffle)
        self.partial = bool(partial)
        self.prefetch = bool(prefetch)
        self.prepare = prepare
        self._rng = np.random.default_rng(seed)
        self._side = None
        self._pending = None          # (perm, perm_dev, first prepared batch) of the next epoch, made while this one's last batch runs

    def __len__(self):
        n = len(self.data)
        return (n + self.batchsize - 1) // self.batchsize if self.partial else n // self.batchsize

    def _draw(self):
        n = len(self.data)
        perm = self._rng.permutation(n) if self.shuffle else np.arange(n)
        perm_dev = torch.from_numpy(perm + self.data.gall.index_base).to(self.data.device)      # one upload per epoch
        return perm, perm_dev

    def _make(self, perm, perm_dev, b):
        n = len(self.data)
        lo, hi = b * self.batchsize, min(n, (b + 1) * self.batchsize)
        ids_dev = perm_dev[lo:hi]
        g = self.data.batch(perm[lo:hi], ids_dev)
        if self.prepare is not None:
            self.prepare(g)
        tgt = self.data.targets_of(ids_dev) if self.data.targets is not None else None
        return g, tgt

    def __iter__(self):
        if not self.prefetch:
            perm, perm_dev = self._draw()
            for b in range(len(self)):
                g, tgt = self._make(perm, perm_dev, b)
                yield (g, tgt) if tgt is not None else g
            return
        dev = self.data.device
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        side = self._side

        def on_side(fn):
            with torch.cuda.stream(side):
                out = fn()
                ev = torch.cuda.Event()
                ev.record(side)
            return out, ev

        def first_of_new_epoch():
            def go():
                perm, perm_dev = self._draw()
                return perm, perm_dev, self._make(perm, perm_dev, 0)
            (perm, perm_dev, first), ev = on_side(go)
            return perm, perm_dev, first, ev

        if self._pending is None:
            self._pending = first_of_new_epoch()
        perm, perm_dev, nxt, ev = self._pending
        self._pending = None
        nb = len(self)
        for b in range(nb):
            (g, tgt), ev_b = nxt, ev
            if b + 1 < nb:          # enqueue the NEXT batch's preparation before this one is handed out
                nxt, ev = on_side(lambda: self._make(perm, perm_dev, b + 1))
            else:                   # ... or the next epoch's first batch
                self._pending = first_of_new_epoch()
            main.wait_event(ev_b)
            for t in (g.x, g.graph_indicator, g._cache.get("node_ptr"), g._cache.get("node_map"), tgt):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
            yield (g, tgt) if tgt is not None else g


def concat_plans(plans, want_indicator=False, idx_dtype=torch.int64, index_base=1):
    """gnnmp_plan_concat over Plan objects (the plan of MLUtils.batch(gs) from the members' cached plans): returns (Plan, seg_ptr[, gi])"""
    k = len(plans)
    dev = plans[0].device
    n_rows = sum(p.n_dst for p in plans)
    arr = (ctypes.c_void_p * k)(*[p.handle for p in plans])
    seg = torch.empty(k + 1, dtype=torch.int64, device=dev)
    gi = torch.empty(n_rows, dtype=idx_dtype, device=dev) if want_indicator else None
    h = ctypes.c_void_p()
    L.check(L.load().gnnmp_plan_concat(ctypes.byref(h), arr, k, L.ptr(seg), L.ptr(gi), 8 if idx_dtype == torch.int64 else 4, index_base,
                                      L.stream_ptr()))
    plan = Plan._adopt(h, dev)
    return (plan, seg, gi) if want_indicator else (plan, seg)
