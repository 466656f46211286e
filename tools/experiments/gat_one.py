#!/usr/bin/env python
"""One-pass attention kernel alone on the products shape (H = 8, C = 16): median of 15 launches.
python tools/experiments/gat_one.py [knob=value ...]   (8 = fast exp)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

lib = L.load()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    lib.gnnmp_tune(int(k), int(v))
N, E = synth.PRODUCTS["N"], synth.PRODUCTS["E"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
H, C = 8, 16
Wx = torch.randn((N, H * C), device="cuda") * 0.3
a = torch.randn((H, 2 * C), device="cuda") * 0.3
b = torch.randn(H * C, device="cuda") * 0.1
out = torch.empty_like(Wx)


def run():
    L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, L.ptr(b), L.ACT_RELU, L.ptr(out), H, C,
                                   L.stream_ptr()))


run(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
for x, y in ev:
    x.record(); run(); y.record()
torch.cuda.synchronize()
ts = sorted(x.elapsed_time(y) for x, y in ev)
Ep = E + N
alg = Ep * (4 * H * C + 4) + N * (8 * H * C + 8)
print(f"knobs {sys.argv[1:]} gat one-pass: median {ts[7]:.3f} ms  min {ts[0]:.3f}  {alg / ts[7] / 1e6:.0f} GB/s alg  checksum {float(out.double().sum()):.6e}")
