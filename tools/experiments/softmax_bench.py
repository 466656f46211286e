#!/usr/bin/env python
"""Standalone softmax_edge_neighbors (GNNlib/src/utils.jl:84-97) on the products / arxiv shapes, H = 8 and H = 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for name, (s, tt), N in (("arxiv", synth.arxiv_like(), synth.ARXIV["N"]), ("products", synth.products_like(), synth.PRODUCTS["N"])):
    g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(tt).cuda(), num_nodes=N, _validated=True)
    g.plan(False)
    E = g.num_edges
    for H in (8, 1):
        e = torch.randn((E, H), device="cuda")
        ms = t(lambda: gnnmp.softmax_edge_neighbors(g, e))
        alg = E * (8 * H + 4)          # read logits once + write alpha once + one index per edge
        print(f"{name}: softmax_edge_neighbors E={E} H={H}: {ms:.3f} ms  ({alg/ms/1e6:.0f} GB/s algorithmic, {E/ms/1e6:.2f} G edges/s)")
