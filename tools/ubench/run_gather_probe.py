#!/usr/bin/env python
"""tools/ubench/gather_probe.hip on the GPU: random-row gather ceiling (G lines/s, TB/s) by row width, loads in flight and matrix
size (HBM-resident 2.4 M rows / Infinity-Cache-resident 170 k rows), uniform ids and the products plan's real column ids."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libgather_probe.so"))
lib.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_void_p, ctypes.c_void_p]
n_ids = 64_308_169


def med(fn, n=7):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


for N in (2_449_029, 169_343):
    ids = torch.randint(0, N, (n_ids,), device="cuda", dtype=torch.int32)
    for D in (128, 100, 32):
        x = torch.randn((N, D), device="cuda")
        log2g = 5 if D > 64 else (4 if D > 32 else 3)
        lines = {128: 4, 100: 4, 32: 1}[D]
        for per_group in (26, 208):
            groups = (n_ids + per_group - 1) // per_group
            out = torch.empty((groups, D), device="cuda")
            for U in (4, 8, 16):
                ms = med(lambda: lib.gather_probe(x.data_ptr(), ids.data_ptr(), n_ids, log2g, D, per_group, U, out.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream))
                print(f"N={N:8d} D={D:3d} rows/group={per_group:3d} U={U:2d}: {ms:6.3f} ms  {n_ids * lines / ms / 1e6:5.1f} G lines/s  "
                      f"{n_ids * D * 4 / ms / 1e9:5.2f} TB/s of rows", flush=True)
        del x


# persistent lane groups: 26 rows per list, waves loop over lists (with / without prefetch of the next list's first ids)
lib.gather_probe_persistent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
N = 2_449_029
ids = torch.randint(0, N, (n_ids,), device="cuda", dtype=torch.int32)
for D in (128, 100):
    x = torch.randn((N, D), device="cuda")
    out = torch.empty(((n_ids + 25) // 26, D), device="cuda")
    for blocks_per_cu in (4, 6, 8, 16):
        for pf in (0, 1):
            ms = med(lambda: lib.gather_probe_persistent(x.data_ptr(), ids.data_ptr(), n_ids, 5, D, 26, pf, 256 * blocks_per_cu,
                                                         out.data_ptr(), torch.cuda.current_stream().cuda_stream))
            print(f"persistent D={D:3d} 26 rows/list blocks/CU={blocks_per_cu:2d} prefetch={pf}: {ms:6.3f} ms  "
                  f"{n_ids * 4 / ms / 1e6:5.1f} G lines/s", flush=True)
    del x, out
