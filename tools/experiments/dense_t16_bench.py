#!/usr/bin/env python
"""dense_t16_kernel (operands straight from HBM, 16x16x4 MFMAs) against the round-1 kernels on the layer shapes of the configs,
with a float64 check of every result.   python tools/experiments/dense_t16_bench.py [waves=...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import gnnmp

sweep = [0]
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "waves":
        sweep = [int(q) for q in v.split(",")]


def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts)//2]


PEAK = 157.3
for (N, K, Dout, two) in [(2449029, 100, 100, False), (2449029, 100, 128, False), (2449029, 100, 256, True),
                          (169343, 128, 128, False), (245246, 16, 128, True), (245246, 128, 128, True), (8192, 128, 4, False),
                          (2708, 64, 64, False), (100000, 52, 36, False), (100000, 24, 200, True)]:
    x = torch.randn((N, K), device="cuda"); m = torch.randn((N, K), device="cuda")
    W = torch.randn((Dout, 2 * K if two else K), device="cuda") * 0.1
    b = torch.randn(Dout, device="cuda")
    f = (lambda: gnnmp.dense(x, W[:, :K], b, "relu", x2=m, W2=W[:, K:])) if two else (lambda: gnnmp.dense(x, W, b, "relu"))
    flops = 2.0 * N * Dout * (2 * K if two else K)
    byts = 4.0 * N * ((2 * K if two else K) + Dout)
    rows = torch.randint(0, N, (2000,), device="cuda")
    xin = torch.cat([x[rows], m[rows]], 1) if two else x[rows]
    ref = torch.relu(xin.double() @ W.double().t() + b.double())
    res = []
    for name, k6 in (("t16", 0), ("r1", 2)):
        gnnmp.tune(6, k6)
        for wv in (sweep if name == "t16" else [0]):
            gnnmp.tune(12, wv)
            y = f()
            err = float((y[rows].double() - ref).abs().max() / ref.abs().max())
            ms = t(f)
            res.append(f"{name}{'' if wv == 0 else f'/w{wv}'} {ms*1e3:7.1f} us {flops/ms/1e9:6.1f} TF ({flops/ms/1e9/PEAK*100:4.1f}%) "
                       f"{byts/ms/1e6:5.0f} GB/s err {err:.1e}")
    gnnmp.tune(6, 0); gnnmp.tune(12, 0)
    print(f"N={N} K={K}{'x2' if two else ''} Dout={Dout}:\n   " + "\n   ".join(res), flush=True)
