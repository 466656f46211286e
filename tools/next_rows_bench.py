#!/usr/bin/env python
"""Measures the SURVEY.md §8f rows that were built after the headline path, on the products shape (N = 2.45 M,
E = 61.9 M, D = 100 unless stated): attention layers sharing the one-pass kernel, message functions, graph-wise helpers,
graph prep and the mini-batch pieces.  Prints one line per row: median ms, algorithmic GB/s where a byte count is
meaningful.  python tools/next_rows_bench.py [--quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch, gnnmp
from gnnmp import synth, sampling as S, utils as U
from gnnmp.layers_attn import AGNNConv, GATv2Conv, GINConv, TransformerConv, attn_conv, ATTN_COS, ATTN_DOT, ATTN_GATV2

quick = "--quick" in sys.argv


def t(fn, it=7, sync_each=False):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def wall(fn, it=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def row(name, ms, nbytes=None, note=""):
    bw = f"{nbytes / ms / 1e6:8.0f} GB/s" if nbytes else " " * 13
    print(f"{name:58s} {ms:9.3f} ms {bw}  {note}", flush=True)


N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, tt = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(tt).cuda(), num_nodes=N, _validated=True)
Ep = E + N
x = torch.randn((N, D), device="cuda")
H, C = 8, 16
HC = H * C
plan_l = g.plan(True)
Q = torch.randn((N, HC), device="cuda") * 0.3; K = torch.randn((N, HC), device="cuda") * 0.3; V = torch.randn((N, HC), device="cuda")
a1 = torch.randn((H, C), device="cuda") * 0.3
print(f"products shape N={N} E={E} E'={Ep}")
row("attention GATv2 logit (one pass), H*C=128", t(lambda: attn_conv(plan_l, ATTN_GATV2, K, Q=Q, a=a1, H=H, C=C)), Ep * (4 * HC + 4) + N * (8 * HC + 8))
row("attention dot-product logit + separate V (Transformer)", t(lambda: attn_conv(plan_l, ATTN_DOT, K, Q=Q, V=V, scale=4.0, H=H, C=C)), Ep * (8 * HC + 4) + N * (8 * HC + 8))
xa = torch.randn((N, 128), device="cuda")
row("attention cosine logit (AGNN), D=128, one head", t(lambda: attn_conv(plan_l, ATTN_COS, xa, scale=1.0, H=1, C=128)), Ep * (4 * 128 + 4) + N * (8 * 128 + 8))
l = GATv2Conv((D, C), "relu", heads=H, seed=1); row("GATv2Conv(100=>16, heads 8) layer", t(lambda: l(g, x)))
l2 = TransformerConv((D, C), heads=H, seed=1); row("TransformerConv(100=>16, heads 8, root weight) layer", t(lambda: l2(g, x)))
l3 = AGNNConv(); row("AGNNConv layer, D=100", t(lambda: l3(g, x)), Ep * (4 * D + 4) + N * (8 * D + 8))
l4 = GINConv(gnnmp.Dense((D, D), "relu", seed=1), 0.1); row("GINConv(Dense(100=>100, relu), 0.1) layer", t(lambda: l4(g, x)))
e = torch.randn((E, D), device="cuda")
row("propagate(e_mul_xj, +) with a matrix e (D, E)", t(lambda: gnnmp.propagate(gnnmp.e_mul_xj, g, "+", xj=x, e=e)), E * (8 * D + 8) + N * (4 * D + 8))
del e
row("apply_edges(xi_dot_xj)", t(lambda: gnnmp.apply_edges(gnnmp.xi_dot_xj, g, xi=x, xj=x)), E * (8 * D + 16 + 4))
if not quick:
    row("apply_edges(xi_sub_xj) (writes (D, E))", t(lambda: gnnmp.apply_edges(gnnmp.xi_sub_xj, g, xi=x, xj=x), it=3), E * (12 * D + 16))
lc = gnnmp.CGConv((D, D), "softplus", residual=True, seed=1)
row("CGConv(100=>100, softplus, residual) layer", t(lambda: lc(g, x)))
fs = torch.randn((N, 2 * D), device="cuda") * 0.3
oc = torch.empty((N, D), device="cuda")
from gnnmp import _lib as L
row("  its one-pass kernel (sigmoid(f) .* softplus(s), +)", t(lambda: L.check(L.load().gnnmp_propagate_cg_f32(
    g.plan(False).handle, L.ptr(fs), L.ptr(fs), None, L.ACT_SOFTPLUS, L.ptr(oc), D, L.stream_ptr()))), E * (8 * D + 4) + N * (12 * D + 8))
del fs, oc
lg = gnnmp.GatedGraphConv(D, 2, seed=1); row("GatedGraphConv(100, 2 layers) layer", t(lambda: lg(g, x)))
ld = gnnmp.DConv((D, D), 2, seed=1); row("DConv(100=>100, k=2) layer", t(lambda: ld(g, x)))
del lc, lg, ld
torch.cuda.empty_cache()
we = torch.randn((E, 64), device="cuda") * 0.1            # nn(e) for an 8 => 8 NNConv: one (8, 8) matrix per edge
x8 = torch.randn((N, 8), device="cuda"); o8 = torch.empty((N, 8), device="cuda")
row("NNConv propagate (8 => 8): per-edge matvec + aggregation", t(lambda: L.check(L.load().gnnmp_propagate_nn_f32(
    g.plan(False).handle, L.SUM, L.ptr(x8), L.ptr(we), L.ptr(o8), 8, 8, L.stream_ptr()))), E * (4 * 64 + 4 * 8 + 8) + N * 4 * 8)
del we, x8, o8
torch.cuda.empty_cache()
eh = torch.randn((E, H), device="cuda")
row("softmax_edge_neighbors, H=8", t(lambda: gnnmp.softmax_edge_neighbors(g, eh)), E * (8 * H + 4))
del eh
torch.cuda.empty_cache()
# graph prep (wall clock: these synchronise)
row("plan_create (dst-sorted CSR of 61.9 M Int64 edges)", wall(lambda: gnnmp.graph.Plan(g.s, g.t, N, N, 1, False, validate=False)))
row("sort_edge_index (61.9 M pairs, 64-bit radix sort)", wall(lambda: S.sort_edge_index(g.s, g.t)), E * 16 * 2)
row("is_bidirected", wall(lambda: S.is_bidirected(g)))
row("has_self_loops", wall(lambda: S.has_self_loops(g)), E * 16)
seeds = torch.from_numpy(np.random.default_rng(0).permutation(N)[:100_000] + 1).cuda()
row("sample_neighbors(100k seeds, K=10)", wall(lambda: S.sample_neighbors(g, seeds, 10, seed=1)))
row("induced_subgraph(100k nodes)", wall(lambda: S.induced_subgraph(g, seeds)))
g.x = x
ld = S.NeighborLoader(g, num_neighbors=[10, 5], num_layers=2, input_nodes=seeds[:4096], batch_size=1024, seed=2)
def one_epoch():
    n = 0
    for mb in ld:
        n += mb.num_nodes
    return n
ms = wall(one_epoch, it=2)
row(f"NeighborLoader [10,5], 4 batches of 1024 seeds ({one_epoch() // 4} nodes/batch)", ms / 4, note="per mini-batch")
# graph-wise helpers on the batched config
members = synth.batched_graphs(G=8192)
gb = gnnmp.batch_arrays(members, [np.random.default_rng(1).standard_normal((n, 16), dtype=np.float32) for _, _, n in members])
xb = gb.x
row("softmax_nodes (8192 graphs, 246k nodes, D=16)", t(lambda: U.softmax_nodes(gb, xb)))
eb = torch.randn((gb.num_edges, 16), device="cuda")
row("softmax_edges (8192 graphs, 983k edges, D=16)", t(lambda: U.softmax_edges(gb, eb)))
row("reduce_edges(mean)", t(lambda: U.reduce_edges("mean", gb, eb)))
