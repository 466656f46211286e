#!/usr/bin/env python
"""The small configs of BASELINE.json alone, for rocprofv3 kernel traces and A/B runs:
    python tools/small_configs.py arxiv   [knob=value ...]   GCNConv(128=>128,relu) and GATConv(128=>16,h=8,relu) forward, arxiv shape
    python tools/small_configs.py batched [knob=value ...]   config 5: 8192 graphs, GraphConv x2 + GlobalPool(mean) + Dense
    python tools/small_configs.py sage [noplace] [knob=value ...]   config 4: SAGEConv(100=>256, relu; mean) forward, products shape
Prints the wall time per layer / step (median of per-call HIP events and the back-to-back average)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch, gnnmp
from gnnmp import synth

what = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
noplace = "noplace" in sys.argv[2:]        # sage: fresh allocations instead of the placement arena (no probe launches in a kernel trace)
for kv in [a for a in sys.argv[2:] if a != "noplace"]:
    k, v = kv.split("=")
    gnnmp.tune(int(k), int(v))


def measure(fn, name, it=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / it * 1e3
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{name}: back-to-back {wall * 1e3:7.1f} us/call, event median {ts[len(ts) // 2] * 1e3:7.1f} us", flush=True)


if what == "arxiv":
    N, D = synth.ARXIV["N"], 128
    s, t = synth.arxiv_like()
    g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
    x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
    gcn = gnnmp.GCNConv((D, D), "relu", seed=11)
    gat = gnnmp.GATConv((D, 16), "relu", heads=8, seed=12)
    p = g.plan(True)
    print("arxiv shape: E' =", p.n_total, "split threshold", p.long_thresh, "split rows", p.n_long, "max degree", p.max_degree)
    measure(lambda: gcn(g, x), "GCNConv(128=>128) layer")
    measure(lambda: gat(g, x), "GATConv(128=>16,h=8) layer")
elif what == "sage":
    N, D = synth.PRODUCTS["N"], synth.PRODUCTS["D"]
    s, t = synth.products_like()
    g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
    x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
    sage = gnnmp.SAGEConv((D, 256), "relu", aggr="mean", seed=13)
    sage.place_outputs = not noplace
    print("products shape: E =", g.num_edges)
    measure(lambda: sage(g, x), "SAGEConv(100=>256, mean) layer", it=30)
else:
    members = synth.batched_graphs(G=8192)
    rng = np.random.default_rng(4)
    gb = gnnmp.batch_arrays(members, [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members])
    gb.plan(False)
    model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                           gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
    print("batched: nodes", gb.num_nodes, "edges", gb.num_edges)
    measure(lambda: model(gb, gb.x), "config-5 step")
    l1, l2, pool, head = model.layers
    h1 = l1(gb, gb.x); h2 = l2(gb, h1); u = pool(gb, h2)
    measure(lambda: l1(gb, gb.x), "  GraphConv(16=>128)")
    measure(lambda: l2(gb, h1), "  GraphConv(128=>128)")
    measure(lambda: pool(gb, h2), "  GlobalPool(mean)")
    measure(lambda: head(u), "  Dense(128=>2)")
