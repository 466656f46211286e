#!/usr/bin/env python
"""ΔW (+ Δb) kernel alone: gnnmp_dense_grad_w_f32 at N x Dout x K (default 2449029 x 100 x 100), median of 11 launches.
usage: gradw_one.py [shape=N,Dout,K] [knob=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L
from gnnmp.backward import dense_grad_w

lib = L.load()
N, Dout, K = 2449029, 100, 100
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "shape":
        N, Dout, K = (int(a) for a in v.split(","))
    else:
        lib.gnnmp_tune(int(k), int(v))
dz = torch.randn((N, Dout), device="cuda"); x = torch.randn((N, K), device="cuda")
for need_b in (True, False):
    fn = lambda: dense_grad_w(dz, x, need_b=need_b)
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(11)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[5]
    print(f"knobs {[a for a in sys.argv[1:] if not a.startswith('shape')]} {N}x{Dout}x{K} dW{'+db' if need_b else ''}: "
          f"{ms * 1e3:.1f} us  {2 * N * Dout * K / ms / 1e9:.1f} TF  ({(N * (Dout + K) * 4) / ms / 1e6:.0f} GB/s)", flush=True)
