#!/usr/bin/env python
"""Third placement map: MANY separately allocated 1.17 GiB buffers (one hipMalloc each through torch's allocator); each timed as the
attention kernel's output against two fixed sources (the first and the last buffer), and a few as sources.  Prints addresses and classes.
    python tools/experiments/placement_map3.py [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

CNT = int(sys.argv[1]) if len(sys.argv) > 1 else 48
lib = L.load()
N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
H, C = 8, 16
HC = H * C
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
gat = gnnmp.GATConv((D, C), "relu", heads=H, seed=12)
Wx0 = gnnmp.dense(x, gat.dense_x_weight)
a_hc = gat.a_hc


def timed(src, dst, reps=5):
    f = lambda: L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(src), None, L.ptr(a_hc), 0.2, L.ptr(gat.bias), L.ACT_RELU, L.ptr(dst), H, C, L.stream_ptr()))
    f(); f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


bufs = [torch.empty((N, HC), device="cuda") for _ in range(CNT)]
for i in (0, CNT // 2, CNT - 1):
    bufs[i].copy_(Wx0)
print("free memory after the allocations: %.1f GiB" % (torch.cuda.mem_get_info()[0] / 2**30))
for si in (0, CNT // 2, CNT - 1):
    row = []
    for j, b in enumerate(bufs):
        if j == si:
            row.append("  -  ")
            continue
        if j in (0, CNT // 2, CNT - 1):        # (a source buffer: do not overwrite it)
            row.append("  s  ")
            continue
        row.append(f"{timed(bufs[si], b):.2f}")
    print(f"source = buffer {si:2d} @ {bufs[si].data_ptr():#x}: " + " ".join(row), flush=True)
print("addresses (GiB, low 40 bits): " + " ".join(f"{(b.data_ptr() & ((1 << 40) - 1)) / 2**30:.2f}" for b in bufs))
