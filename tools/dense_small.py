#!/usr/bin/env python
"""Dense kernel on small-N shapes (arxiv: 169 343 x 128 => 128; batched config: 245 760 x 16/128 => 128):
python tools/dense_small.py [knob=value ...]   (6 = force the K-chunked kernel, 7 = token/skew bits)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L

lib = L.load()
shapes = ((169343, 128, 128), (245760, 16, 128), (245760, 128, 128), (20000, 128, 128), (2449029, 100, 100))
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "shape":                      # shape=N,K,Dout: only this one (PMC runs average per kernel name)
        shapes = (tuple(int(q) for q in v.split(",")),)
    else:
        lib.gnnmp_tune(int(k), int(v))


def t(fn, it=30):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for N, K, Dout in shapes:
    x = torch.randn((N, K), device="cuda"); W = torch.randn((Dout, K), device="cuda") * 0.1; b = torch.randn(Dout, device="cuda")
    ms = t(lambda: gnnmp.dense(x, W, b, "relu"))
    print(f"knobs {sys.argv[1:]} {N}x{K}=>{Dout}: {ms*1e3:8.1f} us  {2*N*K*Dout/ms/1e9:6.1f} TF  ({4*N*(K+Dout)/ms/1e6:5.0f} GB/s)")
