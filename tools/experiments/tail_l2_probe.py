#!/usr/bin/env python
"""VERDICT r5 item 2, feasibility before any kernel is written: a 400-byte row costs four 128-byte line requests.  Split layout
[N][96] (three aligned lines) + [N][4] tail (16 B): what would the tail pass cost (a) gathered at random from the whole 39 MB tail
array, (b) with every XCD confined to a source range whose tails fit its 4 MB L2 (emulated: ids drawn from an N/8- or N/16-row
slice, so EVERY L2 holds the slice — an upper bound on what per-XCD ranges could give)?  Also the body at D = 96 through the
library's own row kernel, and the one-off relayout (a strided copy)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libgather_probe.so"))
lib.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_void_p, ctypes.c_void_p]
n_ids = 64_308_169
N = 2_449_029


def med(fn, n=9):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


x4 = torch.randn((N, 4), device="cuda")
for rows, what in ((N, "whole tail array, 39 MB"), (N // 8, "N/8 slice, 4.9 MB"), (N // 16, "N/16 slice, 2.4 MB"), (N // 64, "N/64 slice, 0.6 MB")):
    ids = torch.randint(0, rows, (n_ids,), device="cuda", dtype=torch.int32)
    for per_group in (26, 208):
        groups = (n_ids + per_group - 1) // per_group
        out = torch.empty((groups, 4), device="cuda")
        for U in (8, 16):
            ms = med(lambda: lib.gather_probe(x4.data_ptr(), ids.data_ptr(), n_ids, 0, 4, per_group, U, out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
            print(f"tail D=4 ids in {what:28s} rows/group={per_group:3d} U={U:2d}: {ms:6.3f} ms  {n_ids / ms / 1e6:6.1f} G rows/s", flush=True)
    del ids

# the relayout [N][100] -> [N][96] + [N][4] (what a producer that cannot write the split layout itself would pay per call)
x = torch.randn((N, 100), device="cuda")
body = torch.empty((N, 96), device="cuda")
tail = torch.empty((N, 4), device="cuda")
def relayout():
    body.copy_(x[:, :96]); tail.copy_(x[:, 96:])
print(f"relayout by two torch strided copies: {med(relayout):.3f} ms", flush=True)
flat = torch.empty_like(x)
print(f"plain copy of the same 0.98 GB: {med(lambda: flat.copy_(x)):.3f} ms", flush=True)

# the body through the library's row kernel on the real plan
import gnnmp
from gnnmp import _lib as L, synth
glib = L.load()
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
for D in (96, 100, 4):
    xx = torch.randn((N, D), device="cuda")
    oo = torch.empty_like(xx)
    ms = med(lambda: L.check(glib.gnnmp_propagate_f32(plan.handle, L.COPY_XJ, L.SUM, L.ptr(xx), None, None, None, L.ptr(oo), D, L.stream_ptr())))
    print(f"csr_rows_kernel on the products plan, D={D:3d}: {ms:.3f} ms", flush=True)
