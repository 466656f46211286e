#!/usr/bin/env python
"""dense_split_kernel experiment variants (library built with EXTRA=-DGNNMP_SPLIT_EXPERIMENTS): knob 13 = VAR (dense_split.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import gnnmp


def t(fn, it=12):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts)//2]


NAMES = {0: "production (Dout%DP!=0 form)", 256: "production", 257: "1-kb ring slots", 258: "no peel", 260: "768/1024 threads", 1280: "ring 3 pairs",
         2304: "ring = whole tile", 2308: "ring = whole tile, 768/1024 thr", 264: "predicated stores", 272: "no stores", 288: "no split arithmetic",
         320: "no A reads", 384: "same rows", 400: "same rows, no stores", 496: "same rows, no stores/split/A reads", 768: "no NaN test",
         2320: "ring = whole tile, no stores"}
for (N, K, Dout, two) in [(2449029, 100, 128, False), (245246, 128, 128, True)]:
    x = torch.randn((N, K), device="cuda"); m = torch.randn((N, K), device="cuda")
    W = torch.randn((Dout, 2 * K if two else K), device="cuda") * 0.1
    b = torch.randn(Dout, device="cuda")
    f = (lambda: gnnmp.dense(x, W[:, :K], b, "relu", x2=m, W2=W[:, K:])) if two else (lambda: gnnmp.dense(x, W, b, "relu"))
    flops = 2.0 * N * Dout * (2 * K if two else K)
    print(f"N={N} K={K}{'x2' if two else ''} Dout={Dout}  ({flops/1e9:.1f} GFLOP)", flush=True)
    gnnmp.tune(17, -1); print(f"   fp32-mfma                                  {t(f)*1e3:8.1f} us"); gnnmp.tune(17, 0)
    for var, name in NAMES.items():
        if var == 0:
            continue
        gnnmp.tune(13, var)
        base = t(f)
        wv = []
        for w in (4, 6):
            gnnmp.tune(12, w)
            wv.append(f"w{w}:{t(f)*1e3:7.1f}")
        gnnmp.tune(12, 0)
        print(f"   VAR{var:5d} {name:38s} {base*1e3:8.1f} us | " + " ".join(wv), flush=True)
    gnnmp.tune(13, 0)
# the other shapes, production kernel vs fp32-MFMA
for (N, K, Dout, two) in [(2449029, 100, 100, False), (2449029, 100, 256, True), (169343, 128, 128, False), (245246, 16, 128, True), (100000, 52, 36, False)]:
    x = torch.randn((N, K), device="cuda"); m = torch.randn((N, K), device="cuda")
    W = torch.randn((Dout, 2 * K if two else K), device="cuda") * 0.1
    b = torch.randn(Dout, device="cuda")
    f = (lambda: gnnmp.dense(x, W[:, :K], b, "relu", x2=m, W2=W[:, K:])) if two else (lambda: gnnmp.dense(x, W, b, "relu"))
    gnnmp.tune(17, -1); t32 = t(f); gnnmp.tune(17, 0)
    print(f"N={N} K={K}{'x2' if two else ''} Dout={Dout}: split {t(f)*1e3:8.1f} us   fp32-mfma {t32*1e3:8.1f} us", flush=True)
