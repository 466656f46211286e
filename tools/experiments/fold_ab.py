"""A/B on one box: split rows folded inside the row kernel (default) vs by the combine kernels (knob 19 bit 7), arxiv and products shape.
usage: python tools/experiments/fold_ab.py [arxiv|products]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth
shape = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
if shape == "arxiv":
    N, D = synth.ARXIV["N"], 128
    s, t = synth.arxiv_like()
    iters = 300
else:
    N, D = synth.PRODUCTS["N"], 100
    s, t = synth.products_like()
    iters = 20
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
gcn = gnnmp.GCNConv((D, D), "relu", seed=11); gat = gnnmp.GATConv((D, 16), "relu", heads=8, seed=12)
p = g.plan(True); print(shape, "thresh", p.long_thresh, "long", p.n_long, "maxdeg", p.max_degree, flush=True)
fns = {"gcn": lambda: gcn(g, x), "gat": lambda: gat(g, x), "propagate": lambda: gnnmp.propagate(gnnmp.copy_xj, g, "+", xj=x)}
base = gnnmp.knob(19)
def timed(f):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
for rep in range(3):
    for name, f in fns.items():
        out = {}
        for label, kv in (("fold", base & ~128), ("two-kernel", base | 128)):
            gnnmp.tune(19, kv)
            out[label] = timed(f)
        print(f"rep {rep} {name:10s} fold {out['fold']:.4f} ms   two-kernel {out['two-kernel']:.4f} ms", flush=True)
gnnmp.tune(19, base)
