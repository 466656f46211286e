#!/usr/bin/env python
"""tools/ubench/mfma_probe.hip on the GPU: TF/s and pipe utilisation of v_mfma_f32_16x16x4_f32 by waves per SIMD, accumulators, LDS feed"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmfma_probe.so"))
lib.mfma_probe.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(1 << 20, device="cuda")
iters = 4000
for nacc in (4, 7, 8):
    for use_lds in (0, 1):
        for waves in (4, 8, 12, 16):
            def run():
                rc = lib.mfma_probe(nacc, use_lds, waves, iters, 256, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            run(); torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in ev:
                a.record(); run(); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev)[2]
            mf = 256 * waves * iters * 4 * nacc                      # MFMAs
            tf = mf * 2 * 16 * 16 * 4 / ms / 1e9
            cyc_per = ms * 1e-3 * 2.4e9 / (mf / 1024)                 # cycles per MFMA per SIMD at 2.4 GHz
            print(f"acc={nacc} lds={use_lds} waves/CU={waves:2d}: {ms:7.3f} ms  {tf:6.1f} TF  ({cyc_per:5.1f} cyc/MFMA/SIMD @2.4GHz)", flush=True)
