#!/usr/bin/env python
"""Chunk length of split rows (GNNMP_CHUNK_SLOTS at plan build: 512 = rounds 1-5, 128 = round 6) on the products shape: the fused GCN layer
(whose pre-pass over the split rows runs alone), the scaled propagate, the attention kernel and SAGE's mean aggregation.  One box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth

N, D = synth.PRODUCTS["N"], synth.PRODUCTS["D"]
s, t = synth.products_like()
sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()


def med(fn, it=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[it // 2]


outs = {}
for rnd in range(2):
    for cs in ("512", "128", "64"):
        os.environ["GNNMP_CHUNK_SLOTS"] = cs
        g = gnnmp.GNNGraph(sd, td, num_nodes=N, _validated=True)
        p = g.plan(True)
        gcn = gnnmp.GCNConv((D, D), "relu", seed=11)
        gat = gnnmp.GATConv((D, 16), "relu", heads=8, seed=12)
        y = gcn(g, x)
        outs.setdefault(cs, y.clone())
        print(f"round {rnd} chunk slots {cs:>4s}: GCN layer {med(lambda: gcn(g, x)):.3f} ms | GAT layer {med(lambda: gat(g, x)):.3f} ms | "
              f"propagate(mean) {med(lambda: gnnmp.propagate(gnnmp.copy_xj, g, 'mean', xj=x)):.3f} ms | max |y - y_512| / max|y| "
              f"{float((y - outs['512']).abs().max() / outs['512'].abs().max()):.1e}", flush=True)
        del g, p
