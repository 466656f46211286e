#!/usr/bin/env python
"""graph_chain2_kernel launch variants (knob 13: low 2 bits = waves per block 0: 8, 1: 4, 2: 6; bit 2 = no scheduling barriers) and the
arxiv-size dense product over waves per block (knob 12)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    return ts[len(ts) // 2]


members = synth.batched_graphs(G=8192)
rng = np.random.default_rng(4)
xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
g = gnnmp.batch_arrays(members, xs)
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
f = lambda: model(g, g.x)
y0 = f()
for kv in (0, 1, 2, 4, 5, 6):
    gnnmp.tune(13, kv)
    y = f()
    print(f"chain2 knob13={kv} (waves {[8,4,6][kv & 3]}, sched barriers {'off' if kv & 4 else 'on'}): {t(f)*1e3:7.1f} us  equal {bool(torch.equal(y, y0))}", flush=True)
gnnmp.tune(13, 0)
for (N, K, Dout) in [(169343, 128, 128), (245246, 128, 128)]:
    x = torch.randn((N, K), device="cuda"); W = torch.randn((Dout, K), device="cuda") * 0.1; b = torch.randn(Dout, device="cuda")
    fd = lambda: gnnmp.dense(x, W, b, "relu")
    row = [f"auto {t(fd)*1e3:6.1f}"]
    for w in (4, 5, 6, 7, 8):
        gnnmp.tune(12, w); row.append(f"w{w} {t(fd)*1e3:6.1f}")
    gnnmp.tune(12, 0)
    print(f"dense {N}x{K}=>{Dout}: " + "  ".join(row), flush=True)
