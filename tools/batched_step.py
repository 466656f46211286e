#!/usr/bin/env python
"""BASELINE.json config 5, the production step only (what bench.py's extras.batched times): 8192 batched graphs through the fused chain,
300 steps — for a kernel trace that shows nothing else.   rocprofv3 --kernel-trace --stats -- python tools/batched_step.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import bench

step, G, nodes, edges = bench.batched_setup(0, 1, None)
for _ in range(300):
    step()
torch.cuda.synchronize()
print("config 5:", G, "graphs,", nodes, "nodes,", edges, "edges: 300 steps")
