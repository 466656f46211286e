#!/usr/bin/env python
"""Per-batch preparation of config 5 (what a training loop pays once per NEW batch, outside bench.py's timed region): the plan of the
batched graph and the chain jobs (gnnmp_chain_jobs_create: device -> host copy of seg_ptr, best-fit-decreasing packing, upload)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gnnmp
from gnnmp import synth, layers

members = synth.batched_graphs(G=8192)
rng = np.random.default_rng(4)
xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
g0 = gnnmp.batch_arrays(members, xs); model(g0, g0.x); torch.cuda.synchronize()      # library warm-up
res = {"batch": [], "plan": [], "jobs + first step": [], "steady step": []}
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = gnnmp.batch_arrays(members, xs); torch.cuda.synchronize(); t1 = time.perf_counter()
    g.plan(False); torch.cuda.synchronize(); t2 = time.perf_counter()
    model(g, g.x); torch.cuda.synchronize(); t3 = time.perf_counter()
    model(g, g.x); torch.cuda.synchronize(); t4 = time.perf_counter()
    for k, v in zip(res, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        res[k].append(v * 1e3)
for k, v in res.items():
    print(f"{k:20s} median {sorted(v)[len(v) // 2]:8.3f} ms")
