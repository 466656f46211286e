#!/usr/bin/env python
"""Float64 propagate (gnnmp_propagate_f64): the products shape at D = 100, and the reference's own micro-benchmark shape
(GraphNeuralNetworks/perf/bench_gnn.jl: n = 1024, density 0.01, B = rand(100, n), propagate(e_mul_xj, g, +))."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, gnnmp
from gnnmp import synth


def med(fn, it=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[it // 2]


rng = np.random.default_rng(0)
n = 1024
m = int(0.01 * n * n)
s = torch.from_numpy(rng.integers(1, n + 1, m)).cuda(); t = torch.from_numpy(rng.integers(1, n + 1, m)).cuda()
g = gnnmp.GNNGraph(s, t, num_nodes=n)
B = torch.rand((n, 100), dtype=torch.float64, device="cuda"); e = torch.rand(m, dtype=torch.float64, device="cuda")
g.plan(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000):
    gnnmp.propagate(gnnmp.e_mul_xj, g, "+", xj=B, e=e)
torch.cuda.synchronize()
print(f"bench_gnn.jl shape (n = 1024, {m} edges, D = 100, Float64): propagate(e_mul_xj, g, +) {(time.perf_counter() - t0) * 1e3:.1f} us per call "
      f"(kernel by events {med(lambda: gnnmp.propagate(gnnmp.e_mul_xj, g, '+', xj=B, e=e)) * 1e3:.1f} us); the reference's comments: ~9 ms generic path, ~400 us B * A on its CPU")

N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
x64 = torch.randn((N, D), dtype=torch.float64, device="cuda")
x32 = x64.to(torch.float32)
for aggr in ("+", "mean", "max"):
    t64 = med(lambda: gnnmp.propagate(gnnmp.copy_xj, g, aggr, xj=x64))
    t32 = med(lambda: gnnmp.propagate(gnnmp.copy_xj, g, aggr, xj=x32))
    b64 = E * (8 * D + 4) + N * (8 * D + 16)
    print(f"products shape, copy_xj {aggr:4s}: Float64 {t64:.3f} ms ({b64 / t64 / 1e6:.0f} GB/s algorithmic; 800-byte rows = 7 lines an edge -> "
          f"{E * 7 / t64 / 1e6:.1f} G lines/s) | Float32 {t32:.3f} ms")
