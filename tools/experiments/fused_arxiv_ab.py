"""arxiv-shaped GCNConv layer: unfused pair (default on cache-resident graphs) vs the fused layer kernel forced (knob 14 = waves cap)"""
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
gcn = gnnmp.GCNConv((D, D), "relu", seed=11)
def timed(f, iters=300):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
ref = gcn(g, x).clone()
variants = sys.argv[1:] or ["0", "16", "12", "10", "8", "6"]
for rep in range(2):
    for v in variants:
        kv = int(v)
        gnnmp.tune(14, kv)
        y = gcn(g, x)
        err = float((y - ref).abs().max())
        print(f"rep {rep} variant {v:>4s}  gcn layer {timed(lambda: gcn(g, x)):.4f} ms   max|diff| vs unfused {err:.2e}", flush=True)
gnnmp.tune(14, 0)
