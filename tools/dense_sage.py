#!/usr/bin/env python
"""SAGEConv(100 => 256)'s contraction [x | m] W^T on the products shape (two column tiles): the grid that runs the column tiles of a
row range side by side (default) against one column tile after the other (knob 13 = 16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import gnnmp


def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for (N, K, Dout) in [(2449029, 100, 256), (245246, 128, 128), (169343, 100, 256)]:
    x = torch.randn((N, K), device="cuda"); m = torch.randn((N, K), device="cuda")
    W = torch.randn((Dout, 2 * K), device="cuda") * 0.1
    b = torch.randn(Dout, device="cuda")
    f = lambda: gnnmp.dense(x, W[:, :K], b, "relu", x2=m, W2=W[:, K:])
    y0 = f()
    row = []
    for kv in (0, 16):
        gnnmp.tune(13, kv)
        y = f()
        row.append(f"knob13={kv}: {t(f)*1e3:8.1f} us equal={bool(torch.equal(y, y0))}")
    gnnmp.tune(13, 0)
    print(f"N={N} K={K}+{K} Dout={Dout}: " + "   ".join(row), flush=True)
