#!/usr/bin/env python
"""Do the placement classes matter at the arxiv size (the gathered matrix fits the Infinity Cache)?  Layers with fresh allocations against
layers with arena buffers (gnnmp.placement, MIN_BYTES = 0), x itself in the arena.
    python tools/experiments/arxiv_placed.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth, placement

Na, Da = synth.ARXIV["N"], synth.ARXIV["D"]
sa, ta = synth.arxiv_like()
ga = gnnmp.GNNGraph(torch.from_numpy(sa).cuda(), torch.from_numpy(ta).cuda(), num_nodes=Na, _validated=True)
xa = torch.from_numpy(synth.features(Na, Da, seed=1)).cuda()
gcn = gnnmp.GCNConv((Da, Da), "relu", seed=11)
gat = gnnmp.GATConv((Da, 16), "relu", heads=8, seed=12)
sage = gnnmp.SAGEConv((Da, Da), "relu", seed=13)


def t(fn, it=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


print(f"fresh allocations: gcn {t(lambda: gcn(ga, xa)):.4f}  gat {t(lambda: gat(ga, xa)):.4f}  sage {t(lambda: sage(ga, xa)):.4f} ms")
placement.MIN_BYTES = 0
ar = placement.arena()
xp = ar.alloc((Na, Da), 0); xp.copy_(xa)
for l in (gcn, gat, sage):
    l.place_outputs = True
print(f"arena buffers:     gcn {t(lambda: gcn(ga, xp)):.4f}  gat {t(lambda: gat(ga, xp)):.4f}  sage {t(lambda: sage(ga, xp)):.4f} ms")
# the worst case on purpose: everything in ONE range
for l in (gcn, gat, sage):
    l.place_outputs = False
import gnnmp.layers as LY
o1 = ar.alloc((Na, Da), 0)
print("(same-class pairs for comparison are whatever torch hands out; see tools/experiments/placement_probe.py for the products shape)")
