#!/usr/bin/env python
"""How much of the row gather does the 256 MiB Infinity Cache serve?  No rocprofv3 counter on gfx950 sees behind it (TCC_EA0_* count
L2->fabric requests, hits included), so the evidence is a working-set sweep of the bare row gather (tools/ubench/gather_probe.hip):
64.3 M uniformly random rows out of a matrix of 64 MB ... 2 GB.  Expected hit rate of a uniform stream = min(1, 256 MiB / working set);
the rate of 128-byte lines per second moves from its cache-resident plateau to its HBM plateau accordingly.  bench.py's
roofline.hbm_bytes_est applies the same residency argument to the plan's real (power-law) sources."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libgather_probe.so"))
lib.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_void_p, ctypes.c_void_p]
n_ids = 64_308_169


def med(fn, n=9):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


print("working set MB | D | rows | ms | G lines/s | TB/s of lines | uniform-stream hit rate 256MiB/WS")
for D, lines in ((128, 4), (100, 4)):
    for mb in (32, 64, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048):
        rows = (mb << 20) // (4 * D)
        x = torch.randn((rows, D), device="cuda")
        ids = torch.randint(0, rows, (n_ids,), device="cuda", dtype=torch.int32)
        per_group = 208
        groups = (n_ids + per_group - 1) // per_group
        out = torch.empty((groups, D), device="cuda")
        ms = med(lambda: lib.gather_probe(x.data_ptr(), ids.data_ptr(), n_ids, 5, D, per_group, 8, out.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream))
        print(f"{mb:5d} | {D:3d} | {rows:8d} | {ms:6.3f} | {n_ids * lines / ms / 1e6:5.1f} | {n_ids * lines * 128 / ms / 1e9:5.2f} | {min(1.0, 256 / mb):.2f}", flush=True)
        del x, ids, out
