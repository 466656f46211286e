#!/usr/bin/env python
"""The one-pass attention kernel runs at 4.90 or 5.15 ms depending on WHICH buffers it reads / writes (tools/experiments/instep_probe.py: same
predecessors, same data).  Here: one 8 GiB arena, the gathered matrix Wx and the output carved out of it at controlled offsets, plus
separately allocated candidates; prints the addresses next to the times (products shape, H = 8, C = 16).
    python tools/experiments/placement_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

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


def run(src, dst):
    L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(src), None, L.ptr(a_hc), 0.2, L.ptr(gat.bias), L.ACT_RELU, L.ptr(dst), H, C, L.stream_ptr()))


def timed(src, dst, reps=8):
    run(src, dst); run(src, dst)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); run(src, dst); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


print("--- separately allocated buffers (torch.empty), every (source, output) pair ---", flush=True)
cands = []
for i in range(4):
    b = torch.empty((N, HC), device="cuda")
    b.copy_(Wx0)
    cands.append(b)
outs = [torch.empty((N, HC), device="cuda") for _ in range(3)]
for i, c in enumerate(cands):
    for j, o in enumerate(outs):
        print(f"src {i} @ {c.data_ptr():#x}  out {j} @ {o.data_ptr():#x}  delta {(o.data_ptr() - c.data_ptr()) / 2**20:10.2f} MiB : {timed(c, o):.3f} ms", flush=True)
ref = outs[0].clone()
del cands, outs
torch.cuda.empty_cache()

print("--- one arena, controlled offsets ---", flush=True)
arena = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
print(f"arena @ {base:#x} (mod 2 MiB = {base % (2 << 20)}, mod 1 GiB = {(base % (1 << 30)) >> 20} MiB)")


def view(off):
    return arena[off: off + nbytes].view(torch.float32).view(N, HC)


GiB = 1 << 30
for so in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096):
    src = view(so)
    src.copy_(Wx0)
    for do in (2 * GiB, 2 * GiB + 256, 2 * GiB + 4096, 2 * GiB + 65536, 2 * GiB + (1 << 20), 2 * GiB + (1 << 21), 3 * GiB, 4 * GiB, 4 * GiB + 4096 * 33):
        dst = view(do)
        tm = timed(src, dst)
        ok = bool(torch.equal(dst, ref))
        print(f"src +{so:9d}  out +{do / GiB:8.5f} GiB  (delta mod 4 KiB {(do - so) % 4096:5d}, mod 2 MiB {((do - so) % (2 << 20)) >> 10:5d} KiB): {tm:.3f} ms {'' if ok else 'MISMATCH'}", flush=True)
