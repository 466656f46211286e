#!/usr/bin/env python
"""Fourth placement map (one 200 GiB arena = buddy blocks of 128 + 64 + 8 GiB: same block or same 64 GiB region?) — derived from the second: is the fast / slow class of (gathered source, output) a matter of ABSOLUTE 32 GiB regions or of the DISTANCE
between the two buffers?  Attention kernel, one 72 GiB arena, source at +8, +20, +36 and +50 GiB, output slid in 2 GiB steps; and the
same question for two arenas allocated separately.
    python tools/experiments/placement_map2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

AG = 200
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
nbytes = N * HC * 4
GiB = 1 << 30
arena = torch.empty(AG * GiB, dtype=torch.uint8, device="cuda")
print(f"arena {AG} GiB @ {arena.data_ptr():#x}", flush=True)


def view(buf, off):
    return buf[off: off + nbytes].view(torch.float32).view(-1, HC)


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


for so in (8, 72, 100, 136, 196):
    src = view(arena, so * GiB); src.copy_(Wx0)
    row = []
    for o in range(0, AG - 2, 4):
        if abs(o - so) < 2:
            continue
        row.append((o, timed(src, view(arena, o * GiB))))
    print(f"source at +{so} GiB, output at: " + " ".join(f"{o}:{tm:.2f}" for o, tm in row), flush=True)
