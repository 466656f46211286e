#!/usr/bin/env python
"""cycles per bf16 MFMA and per SIMD (tools/ubench/bf16_probe.hip): dependent chains vs independent accumulators, 1-3 waves per SIMD"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libbf16_probe.so"))
lib.bf16_probe.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
out = torch.empty(256 * 1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
CLK = 2.4e9
for shape, flop in ((32, 32 * 32 * 16 * 2), (16, 16 * 16 * 32 * 2)):
    for nacc, chain in ((1, 6), (2, 6), (4, 6), (8, 6), (2, 1), (4, 1), (6, 1), (8, 1)):
        row = []
        for waves in (4, 8, 12, 16):
            iters = 2000
            per = lib.bf16_probe(shape, nacc, chain, waves, 10, 256, out.data_ptr(), st)
            if per < 0:
                break
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); lib.bf16_probe(shape, nacc, chain, waves, iters, 256, out.data_ptr(), st); b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            n_mfma_simd = per * iters * waves / 4          # MFMAs issued on one SIMD
            cyc = ms * 1e-3 * CLK / n_mfma_simd
            tf = per * iters * waves * 256 * flop / (ms * 1e-3) / 1e12
            row.append(f"{waves // 4}w/SIMD: {cyc:5.1f} cyc/MFMA {tf:6.0f} TF")
        if row:
            print(f"{shape}x{shape} nacc={nacc} chain={chain}: " + " | ".join(row), flush=True)
