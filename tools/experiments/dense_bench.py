#!/usr/bin/env python
"""Times gnnmp_dense_f32 (W-resident vs K-chunked kernel) on the layer shapes of the bench configs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import gnnmp

def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts)//2]

for (N, K, Dout, two) in [(2449029, 100, 100, False), (2449029, 100, 128, False), (2449029, 100, 256, True),
                          (169343, 128, 128, False), (245760, 16, 128, True), (245760, 128, 128, True)]:
    x = torch.randn((N, K), device="cuda"); m = torch.randn((N, K), device="cuda")
    W = torch.randn((Dout, 2 * K if two else K), device="cuda") * 0.1
    b = torch.randn(Dout, device="cuda")
    f = (lambda: gnnmp.dense(x, W[:, :K], b, "relu", x2=m, W2=W[:, K:])) if two else (lambda: gnnmp.dense(x, W, b, "relu"))
    flops = 2.0 * N * Dout * (2 * K if two else K)
    byts = 4.0 * N * ((2 * K if two else K) + Dout)
    res = []
    for name, k6, k7 in (("wlds+pf", 0, 1), ("wlds", 0, 0), ("chunk", 1, 1)):
        gnnmp.tune(6, k6)
        gnnmp.tune(7, k7)
        ms = t(f)
        res.append(f"{name} {ms:7.3f} ms {flops/ms/1e9:6.1f} TF {byts/ms/1e6:6.0f} GB/s")
    gnnmp.tune(6, 0)
    gnnmp.tune(7, 1)
    print(f"N={N} K={K}{'x2' if two else ''} Dout={Dout}: " + " | ".join(res))
