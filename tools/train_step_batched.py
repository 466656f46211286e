#!/usr/bin/env python
"""The TRAINING step of BASELINE.json config 5 alone (for kernel traces): GNNChain(GraphConv(16 => 128, relu), GraphConv(128 => 128, relu),
GlobalPool(mean), Dense(128 => 2)) on the batch of 8192 synthetic graphs — forward with stored activations, cross-entropy, every pullback
a libgnnmp adjoint kernel (gnnmp/backward.py), Adam.  Prints the wall time per step, eager and replayed from a captured HIP graph.
    python tools/train_step_batched.py [G] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.nn.functional as F
import gnnmp
from gnnmp import synth
from gnnmp.backward import dense_ad, global_pool_ad, graph_chain_ad, graph_conv_ad

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
members = synth.batched_graphs(G=G)
rng = np.random.default_rng(4)
g = gnnmp.batch_arrays(members, [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members])
g.plan(False)
c1, c2 = gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22)
pool, head = gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23)
params = [c1.weight1, c1.weight2, c1.bias, c2.weight1, c2.weight2, c2.bias, head.weight, head.bias]
for p in params:
    p.requires_grad_(True)
Y = torch.from_numpy(rng.integers(0, 2, G)).cuda()
model = gnnmp.GNNChain(c1, c2, pool, head)
LAYERS = "--layers" in sys.argv        # the round-4 path: layer-by-layer adjoints
fwd = (lambda: dense_ad(head, global_pool_ad(pool, g, graph_conv_ad(c2, g, graph_conv_ad(c1, g, g.x))))) if LAYERS else (lambda: graph_chain_ad(model, g, g.x))


def run(opt, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        lg = fwd()
        F.cross_entropy(lg, Y).backward()
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


opt = torch.optim.Adam(params, lr=1e-3, fused=True)
run(opt, 5)
print(f"{'layer-by-layer adjoints' if LAYERS else 'chain pullback'} eager: {run(opt, steps):.3f} ms/step", flush=True)

if "--graph" in sys.argv:
    # the whole step as ONE captured HIP graph (static batch: same pointers every replay); Adam in its capturable form
    opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            lg = fwd()
            F.cross_entropy(lg, Y).backward()
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    cg = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(cg):
        lg = fwd()
        loss = F.cross_entropy(lg, Y)
        loss.backward()
        opt.step()
    for _ in range(3):
        cg.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        cg.replay()
    torch.cuda.synchronize()
    print(f"one HIP graph per step: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step   loss {float(loss):.4f}", flush=True)
