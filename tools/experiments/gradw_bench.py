#!/usr/bin/env python
"""ΔW = Δzᵀ·x (+ Δb) on the products shape: python tools/experiments/gradw_bench.py [knob=value ...]  (9 = slabs per CU, 10 = row pairs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L
from gnnmp.backward import dense_grad_w

lib = L.load()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    lib.gnnmp_tune(int(k), int(v))


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


N = 2_449_029
for K, Dout in ((100, 100), (100, 128), (128, 128), (200, 256)):
    x = torch.randn((N, K), device="cuda"); dz = torch.randn((N, Dout), device="cuda")
    ms = t(lambda: dense_grad_w(dz, x, need_b=False))
    mb = t(lambda: dense_grad_w(dz, x, need_w=False))
    ref = (dz[:200000].double().t() @ x[:200000].double())
    got = dense_grad_w(dz[:200000].contiguous(), x[:200000].contiguous(), need_b=False)[0].double()
    err = float((got - ref).norm() / ref.norm())
    print(f"knobs {sys.argv[1:]} {N}x{K} -> {Dout}: dW {ms:.3f} ms = {2*N*K*Dout/ms/1e9:.1f} TF ({4*N*(K+Dout)/ms/1e6:.0f} GB/s) | db {mb:.3f} ms | rel err {err:.1e}")
