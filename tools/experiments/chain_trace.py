#!/usr/bin/env python
"""Cycle stamps of block (0, 0) of graph_chain2_kernel (library built with EXTRA=-DGNNMP_CHAIN_TRACE): per wave and job, the cycles
spent in each phase."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gnnmp
from gnnmp import synth, _lib as L

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
members = synth.batched_graphs(G=G)
rng = np.random.default_rng(4)
xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
g = gnnmp.batch_arrays(members, xs)
for _ in range(5):
    y = model(g, g.x)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (16 * 8 * 8))()
lib = L.load()
lib.gnnmp_chain_trace.argtypes = [ctypes.c_void_p]
assert lib.gnnmp_chain_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(16, 8, 8)
t0 = t[t > 0].min()
for w in range(12):
    for k in range(8):
        r = t[w, k]
        if r[0] == 0:
            continue
        # stamps: 0 job start, 1 operands of layer 1 formed, 2 K loop done, 3 neighbour rounds done, 4 sigma2 / head done, 5 z rows stored,
        # 6 end of the job (odd waves of one-tile jobs: the next job's prefetch)
        d = [int(r[i + 1] - r[i]) for i in range(6)]
        print(f"wave {w:2d} job {k}: start {int(r[0] - t0):8d}  prologue {d[0]:6d}  K loop {d[1]:6d}  rounds {d[2]:6d}  sigma2/head {d[3]:6d}  "
              f"z store {d[4]:6d}  tail {d[5]:6d}   total {int(r[6] - r[0]):7d}")
