#!/usr/bin/env python
"""GCNConv(100 => 100, relu) layer on the products shape, fused (default) or not: python tools/experiments/gcn_layer_one.py [knob=value ...]
(14 = -1: never fuse; 15 = 0: rows by index instead of by decreasing length)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    gnnmp.tune(int(k), int(v))
N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
x = torch.randn((N, D), device="cuda")
gcn = gnnmp.GCNConv((D, D), "relu", seed=1)
sage = gnnmp.SAGEConv((D, 128), "relu", seed=2)
for name, f in (("GCNConv(100=>100)", lambda: gcn(g, x)), ("SAGEConv(100=>128, mean)", lambda: sage(g, x))):
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"knobs {sys.argv[1:]} {name} layer: median {ts[7]:.3f} ms  min {ts[0]:.3f}")
