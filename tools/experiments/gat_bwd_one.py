#!/usr/bin/env python
"""GATConv attention adjoint on the products shape, alone (for rocprofv3 / knob sweeps):
    python tools/experiments/gat_bwd_one.py [knob=value ...]      e.g.  2=4 (unroll 4)  5=4 (4-wave blocks)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth, backward as bw, _lib as L

lib = L.load()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    lib.gnnmp_tune(int(k), int(v))
N, E = synth.PRODUCTS["N"], synth.PRODUCTS["E"]
s, tt = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(tt).cuda(), num_nodes=N, _validated=True)
H, C = 8, 16
plan, plan_t = g.plan(True), bw.plan_transposed(g, True)
Wx = torch.randn((N, H * C), device="cuda") * 0.3
a_hc = torch.randn((H, 2 * C), device="cuda") * 0.3
dy = torch.randn((N, H * C), device="cuda")
out = torch.empty((N, H * C), device="cuda"); stats = torch.empty((N, H, 2), device="cuda")
line = torch.empty((N, H, 4), device="cuda"); dsd = torch.empty((N, H), device="cuda"); dss = torch.empty((N, H), device="cuda")
dWx = torch.empty((N, H * C), device="cuda"); da = torch.empty((H, 2 * C), device="cuda")
L.check(lib.gnnmp_gat_conv_stats_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, None, 0, L.ptr(out), L.ptr(stats), H, C, L.stream_ptr()))

def run():
    L.check(lib.gnnmp_gat_conv_grad_f32(plan.handle, plan_t.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, L.ptr(stats), L.ptr(dy),
                                        L.ptr(line), L.ptr(dsd), L.ptr(dss), L.ptr(dWx), None, L.ptr(da), H, C, L.stream_ptr()))

for _ in range(3):
    run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); run(); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print(f"knobs {sys.argv[1:]}: gat adjoint median {ts[5]:.3f} ms  min {ts[0]:.3f} ms")
