import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth
N, D = synth.ARXIV["N"], 128
s, t = synth.arxiv_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
gcn = gnnmp.GCNConv((D, D), "relu", seed=11); gat = gnnmp.GATConv((D, 16), "relu", heads=8, seed=12)
for f, name in ((lambda: gcn(g, x), "gcn"), (lambda: gat(g, x), "gat")):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): f()
    torch.cuda.synchronize(); print(name, "layer ms", (time.perf_counter() - t0) * 10)
p = g.plan(True); print("thresh", p.long_thresh, "long", p.n_long, "maxdeg", p.max_degree)
