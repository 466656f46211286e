#!/usr/bin/env python
"""With placement on (arena buffers): the attention / GCN kernels' durations (gnnmp._lib.EventProbe) when the layers run back to back and
when they alternate as in the bench step; classes of every buffer involved.
    python tools/experiments/placed_instep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth, placement

N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
g.plan(True)
gcn = gnnmp.GCNConv((D, D), "relu", seed=11)
gat = gnnmp.GATConv((D, 16), "relu", heads=8, seed=12)
gcn.place_outputs = gat.place_outputs = True
gcn(g, x); gat(g, x)
ar = placement.arena()
print("arena", ar.info(), "class of x", ar.class_of(x))
for lname, l in (("gcn", gcn), ("gat", gat)):
    for k, b in l._placed.items():
        print(f"  {lname} buffer {k[0]} in range {ar.class_of(b)} @ {b.data_ptr():#x}")


def run(name, fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    pr = L.EventProbe()
    L.set_probe(pr)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    L.set_probe(None)
    out = []
    for k in ("fused_conv", "dense", "gat_conv"):
        ts = sorted(pr.times_ms(k))
        if ts:
            out.append(f"{k} {ts[len(ts) // 2]:.3f}")
    print(f"{name:40s} " + "  ".join(out), flush=True)


run("gat layers back to back", lambda: gat(g, x))
run("gcn layers back to back", lambda: gcn(g, x))
run("step: gcn, gat", lambda: (gcn(g, x), gat(g, x)))
run("step: gat, gcn", lambda: (gat(g, x), gcn(g, x)))
x2 = ar.alloc((N, D), 0)
if x2 is not None:
    x2.copy_(x)
    gcn2 = gnnmp.GCNConv((D, D), "relu", seed=11); gat2 = gnnmp.GATConv((D, 16), "relu", heads=8, seed=12)
    gcn2.place_outputs = gat2.place_outputs = True
    gcn2(g, x2); gat2(g, x2)
    for lname, l in (("gcn2", gcn2), ("gat2", gat2)):
        for k, b in l._placed.items():
            print(f"  {lname} buffer {k[0]} in range {ar.class_of(b)}")
    run("x in arena range 0: step gcn, gat", lambda: (gcn2(g, x2), gat2(g, x2)))
    run("x in arena range 0: gat back to back", lambda: gat2(g, x2))
