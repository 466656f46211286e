#!/usr/bin/env python
"""How much of the GCN propagate's time is the 128-byte line straddle of 400-byte rows?  The same products-shaped plan with
D = 96 (384-byte rows = exactly three lines), D = 100, D = 128, and D = 4 (the 16-byte tails alone, a 39 MB array)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

lib = L.load()
N, E = synth.PRODUCTS["N"], synth.PRODUCTS["E"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
Ep = E + N
for D in (96, 100, 128, 4, 32, 64):
    x = torch.randn((N, D), device="cuda")
    out = torch.empty_like(x)
    def run():
        L.check(lib.gnnmp_propagate_f32(plan.handle, L.COPY_XJ, L.SUM, L.ptr(x), None, None, None, L.ptr(out), D, L.stream_ptr()))
    run(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(11)]
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"D={D:4d} ({4*D:4d} B rows, base % 128 = {x.data_ptr() % 128}): median {ts[5]:.3f} ms  {Ep * 4 * D / ts[5] / 1e6:6.0f} GB/s of row bytes", flush=True)
