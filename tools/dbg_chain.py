import os, sys, faulthandler
faulthandler.enable()
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/graphneuralnetworks.jl_amd"); sys.path.insert(0, ROOT + "/tests")
import numpy as np, torch
import gnnmp as gm
gm.load()
import test_graph_chain as T
from gnnmp import synth

def sync(tag):
    torch.cuda.synchronize(); print("ok", tag, flush=True)

for G in (64, 512, 8192):
    members = synth.batched_graphs(G=G, seed=9)
    rng = np.random.default_rng(1)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = T.build(gm, (16, 128, 128), 2, "+", "mean")
    y = model(g, g.x); sync(f"fused G={G}")
    gm.tune(18, -1); yl = model(g, g.x); gm.tune(18, 0); sync(f"layers G={G}")
for seed in (0, 1):
    for case in T.CASES:
        dims, nout, aggr, pool, sigma, bias = case
        rng = np.random.default_rng(100 + seed)
        members = T.random_members(150, rng)
        xs = [rng.standard_normal((n, dims[0]), dtype=np.float32) for _, _, n in members]
        g = gm.batch_arrays(members, xs); sync(f"batch {seed} {dims}")
        model = T.build(gm, dims, nout, aggr, pool, sigma, bias, seed=31 + seed)
        print("maxn", max(n for _,_,n in members), "N", sum(n for _,_,n in members), flush=True)
        if os.environ.get("GENERAL"): gm.tune(18, 1)
        y = model(g, g.x); sync(f"fused {seed} {dims} maxn={max(n for _,_,n in members)}")
        gm.tune(18, -1); yl = model(g, g.x); gm.tune(18, 0); sync(f"layers {seed} {dims}")
