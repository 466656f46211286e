#!/usr/bin/env python
"""hipGraph capture of a launch-bound forward (the batched config-5 model, and the arxiv GCN+GAT pair): the step's libgnnmp
launches are recorded once on a capture stream (torch.cuda.CUDAGraph — plumbing: it owns the stream and the graph object)
and replayed with one host call.  Prints eager vs replayed time per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch, gnnmp
from gnnmp import synth


def t(fn, it=30):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def capture(fn, warm=3):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):            # plans, workspaces and caches are built outside the capture
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


members = synth.batched_graphs(G=8192)
rng = np.random.default_rng(4)
gb = gnnmp.batch_arrays(members, [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members])
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
step = lambda: model(gb, gb.x)
ref = step().clone()
eager = t(step)
graph, out = capture(step)
graph.replay(); torch.cuda.synchronize()
assert torch.equal(out, ref), "replayed output differs"
rep = t(graph.replay)
print(f"batched config 5 (8192 graphs): eager {eager*1e3:.0f} us/step, hipGraph replay {rep*1e3:.0f} us/step ({8192/rep/1e3:.1f} M graphs/s)")

N = synth.ARXIV["N"]
s, tt = synth.arxiv_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(tt).cuda(), num_nodes=N, _validated=True)
x = torch.from_numpy(synth.features(N, 128, seed=1)).cuda()
gcn = gnnmp.GCNConv((128, 128), "relu", seed=1)
gat = gnnmp.GATConv((128, 16), "relu", heads=8, seed=2)
step2 = lambda: (gcn(g, x), gat(g, x))
ref2 = [v.clone() for v in step2()]
eager2 = t(step2)
graph2, out2 = capture(step2)
graph2.replay(); torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(out2, ref2))
rep2 = t(graph2.replay)
Ep = len(s) + N
print(f"arxiv GCNConv + GATConv: eager {eager2*1e3:.0f} us/step, hipGraph replay {rep2*1e3:.0f} us/step ({2*Ep/rep2/1e6:.2f} G edges/s)")
