#!/usr/bin/env python
"""GCN propagate kernel alone on the products shape (D = 100, slot-ordered coefficients): median of 15 launches."""
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
N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
x = torch.randn((N, D), device="cuda")
gcn = gnnmp.GCNConv((D, D), "relu", seed=1)
gcn(g, x)
cvec, c_slot, _ = g._cache[("gcn_norm", True, False)]
out = torch.empty_like(x)


def run():
    L.check(lib.gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(x), None, L.ptr(c_slot), L.ptr(cvec), L.ptr(out), D,
                                          L.stream_ptr()))


run(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
for a, b in ev:
    a.record(); run(); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
Ep = E + N
alg = Ep * (4 * D + 8) + N * (4 * D + 12)
print(f"knobs {sys.argv[1:]} csr propagate: median {ts[7]:.3f} ms  min {ts[0]:.3f}  {alg / ts[7] / 1e6:.0f} GB/s alg")
