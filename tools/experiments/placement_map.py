#!/usr/bin/env python
"""Map the placement classes found by tools/experiments/placement_probe.py: one big arena, the attention kernel's source fixed, its output slid
through the arena in 512 MiB steps (then the other way round); the same for a plain device copy and for the GCN layer kernel.
    python tools/experiments/placement_map.py [arena_GiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth
from gnnmp.layers import gcn_norm_cache

AG = int(sys.argv[1]) if len(sys.argv) > 1 else 40
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
print(f"arena {AG} GiB @ {arena.data_ptr():#x}; plan / x / Wx0 @ {x.data_ptr():#x} {Wx0.data_ptr():#x}", flush=True)


def view(off, n=nbytes, cols=HC):
    return arena[off: off + n].view(torch.float32).view(-1, cols)


def timed(fn, reps=6):
    fn(); fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def gatk(src, dst):
    return lambda: L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(src), None, L.ptr(a_hc), 0.2, L.ptr(gat.bias), L.ACT_RELU, L.ptr(dst), H, C, L.stream_ptr()))


step = GiB // 2
print("--- attention kernel: source at +0, output slid ---", flush=True)
src = view(0); src.copy_(Wx0)
row = []
for k in range(3, 2 * AG - 3):
    row.append((k * 0.5, timed(gatk(src, view(k * step)))))
print(" ".join(f"{o:.1f}:{tm:.2f}" for o, tm in row), flush=True)
print("--- attention kernel: output at +0, source slid ---", flush=True)
dst = view(0)
row = []
for k in range(3, 2 * AG - 3, 2):
    sv = view(k * step); sv.copy_(Wx0)
    row.append((k * 0.5, timed(gatk(sv, dst))))
print(" ".join(f"{o:.1f}:{tm:.2f}" for o, tm in row), flush=True)
print("--- plain copy of 1 GiB: source at +0, destination slid (torch copy_) ---", flush=True)
a1 = arena[:GiB]
row = []
for k in range(3, 2 * AG - 2):
    b1 = arena[k * step: k * step + GiB]
    row.append((k * 0.5, timed(lambda: b1.copy_(a1))))
print(" ".join(f"{o:.1f}:{tm * 1e3:.0f}us" for o, tm in row), flush=True)
print("--- GCN layer kernel: x at +0, output slid ---", flush=True)
gcn = gnnmp.GCNConv((D, D), "relu", seed=11)
cvec, c_slot, _ = gcn_norm_cache(g, True, None)
xb = N * D * 4
xv = view(0, xb, D); xv.copy_(x)
import ctypes
row = []
agg_dummy = None
for k in range(3, 2 * AG - 3, 2):
    ov = view(k * step, xb, D)
    def f(ov=ov):
        # gnnmp_fused_conv_f32 with the caller's output buffer
        y = gnnmp.fused_conv(plan, L.SUM, xv, gcn.weight, gcn.bias, "relu", ss_slot=c_slot, scale_dst=cvec, out=ov)
        assert y is not None
    row.append((k * 0.5, timed(f)))
print(" ".join(f"{o:.1f}:{tm:.2f}" for o, tm in row), flush=True)
