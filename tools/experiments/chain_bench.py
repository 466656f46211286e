#!/usr/bin/env python
"""BASELINE.json config 5 (8192 graphs, GraphConv 16=>128=>128, mean pool, Dense 128=>2): the fused chain kernel against the
layer-by-layer path (knob 18 = -1) and the round-2 dense kernels (knob 17 = -1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gnnmp
from gnnmp import synth


def t(fn, it=50):
    fn(); fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
members = synth.batched_graphs(G=G)
rng = np.random.default_rng(4)
xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
g = gnnmp.batch_arrays(members, xs)
g.plan(False)
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
f = lambda: model(g, g.x)
y = f()
for name, knobs in (("fused chain kernel", ()), ("layer by layer, split-bf16 dense", ((18, -1),)),
                    ("layer by layer, fp32-MFMA dense (round 2)", ((18, -1), (17, -1)))):
    for k, v in knobs:
        gnnmp.tune(k, v)
    y2 = f()
    med, best = t(f)
    err = float((y2 - y).abs().max() / y.abs().max())
    print(f"{name:45s} median {med*1e3:7.1f} us  best {best*1e3:7.1f} us   max |diff to fused| / max = {err:.1e}", flush=True)
    for k, v in knobs:
        gnnmp.tune(k, 0)
# wall-clock per step including the host side (what bench.py's extras.batched reports)
import time
for _ in range(5):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    f()
torch.cuda.synchronize()
print(f"fused, wall clock incl. host: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us/step")
