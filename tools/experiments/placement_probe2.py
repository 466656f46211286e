#!/usr/bin/env python
"""Is there a CHEAP probe that tells the two placement classes apart?  72 GiB arena: output at +20 GiB (same class as the source at +8 GiB:
slow for the attention kernel) against output at +66 GiB (fast).  Probes: gnnmp_gather_f32 of n random 512-byte rows, a streaming copy, the
attention kernel itself on row-range plans of different sizes.
    python tools/experiments/placement_probe2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

lib = L.load()
N, HC = synth.PRODUCTS["N"], 128
GiB = 1 << 30
nbytes = N * HC * 4
arena = torch.empty(72 * GiB, dtype=torch.uint8, device="cuda")


def view(off, rows=N):
    return arena[off: off + rows * HC * 4].view(torch.float32).view(rows, HC)


src = view(8 * GiB)
src.normal_()
outs = {"same (+20 GiB)": 20 * GiB, "other (+66 GiB)": 66 * GiB, "same (+40 GiB)": 40 * GiB, "other (+68 GiB)": 68 * GiB}


def timed(f, reps=9):
    f(); f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3


for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22):
    idx = torch.randint(0, N, (n,), device="cuda", dtype=torch.int32)
    line = []
    for name, off in outs.items():
        dst = view(off, n)
        f = lambda: L.check(lib.gnnmp_gather_f32(L.ptr(src), L.ptr(idx), 4, 0, n, L.ptr(dst), HC, L.stream_ptr()))
        line.append(f"{name} {timed(f):8.1f} us")
    print(f"gather of {n:8d} random 512-byte rows -> " + " | ".join(line), flush=True)
for mb in (64, 256, 1024):
    line = []
    a = arena[8 * GiB: 8 * GiB + (mb << 20)]
    for name, off in outs.items():
        b = arena[off: off + (mb << 20)]
        line.append(f"{name} {timed(lambda: b.copy_(a)):8.1f} us")
    print(f"copy of {mb:5d} MiB -> " + " | ".join(line), flush=True)
# read-only and write-only streams
for name, off in list(outs.items()) + [("source itself (+8 GiB)", 8 * GiB)]:
    b = arena[off: off + GiB]
    w = timed(lambda: b.fill_(1))
    r = timed(lambda: b.view(torch.float32).sum())
    print(f"1 GiB at {name}: fill {w:8.1f} us   sum {r:8.1f} us", flush=True)
