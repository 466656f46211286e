#!/usr/bin/env python
"""Config 5 with a NEW batch every step (the reference's loop): DataLoader on one stream against DataLoader(prefetch = True, prepare =
chain_prepare) — batch k + 1's plan, features and wave jobs on a side stream beside batch k's step.  python tools/experiments/newbatch_prefetch.py [G]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, gnnmp
from gnnmp import synth
from gnnmp.layers import chain_prepare

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
members = synth.batched_graphs(G=G)
rng = np.random.default_rng(4)
xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
ds = gnnmp.GraphDataset.from_members(members, xs)
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))


def run(loader, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for g in loader:
            model(g, g.x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for bs, what in ((G, "one batch = all graphs"), (G // 8, "eight batches an epoch")):
    a = gnnmp.DataLoader(ds, batchsize=bs, shuffle=True, seed=7)
    b = gnnmp.DataLoader(ds, batchsize=bs, shuffle=True, seed=7, prefetch=True, prepare=chain_prepare)
    for r in range(3):
        run(a, 5); ta = run(a, 50) / len(a)
        run(b, 5); tb = run(b, 50) / len(b)
        print(f"{what:24s} round {r}: one stream {ta * 1e3:7.1f} us per step | prefetch on a side stream {tb * 1e3:7.1f} us per step", flush=True)
g = next(iter(gnnmp.DataLoader(ds, batchsize=G)))
model(g, g.x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300):
    model(g, g.x)
torch.cuda.synchronize()
print(f"the step alone on a fixed batch: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us")
