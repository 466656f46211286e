#!/usr/bin/env python
"""config 5's chain kernel against the number of member graphs: the time steps with the number of jobs a wave pair draws (ceil(jobs / 768)
per column slab: 6 pairs a block, 128 blocks a slab), not with the work.  usage: python tools/experiments/chain_rounds.py [G ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch, gnnmp
from gnnmp import synth
Gs = [int(a) for a in sys.argv[1:]] or [7000, 7600, 7900, 8000, 8050, 8100, 8192, 8500, 9000, 9600]
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
allm = synth.batched_graphs(G=max(Gs))
rng = np.random.default_rng(4)
allx = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in allm]
for G in Gs:
    gb = gnnmp.batch_arrays(allm[:G], allx[:G])
    gb.plan(False)
    for _ in range(10): model(gb, gb.x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): model(gb, gb.x)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 300 * 1e3
    jobs = gnnmp.layers.ChainJobs(gb._cache["node_ptr"] if "node_ptr" in gb._cache else gb.node_ptr(), gb.num_graphs) if False else None
    nodes = gb.num_nodes
    print(f"G = {G:5d}  nodes {nodes:7d}  rows/64 = {nodes / 64:7.1f} (>= jobs x 0.98)  per pair >= {nodes / 64 / 768:.2f}   step {ms * 1e3:6.1f} us   {ms * 1e3 / nodes * 1e3:.3f} ns/node", flush=True)
