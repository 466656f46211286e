#!/usr/bin/env python
"""dense_wreg_kernel (W planes resident in registers, Dout = 256) against dense_split_kernel on one box: knob 19 bit 6 (64) turns the
register-resident kernel off.  Prints the median times interleaved and the largest difference between the two results."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import gnnmp


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


SHAPES = [(2449029, 100, 100), (2449029, 64, 64), (2449029, 64, 100), (2449029, 128, 64), (2449029, 64, 0), (2449029, 100, 0),
          (2449029, 128, 0), (2449029, 200, 0), (169343, 128, 0), (169343, 64, 64), (5000, 100, 100)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (N, K1, K2) in SHAPES:
    two = K2 > 0
    x = torch.randn((N, K1), device="cuda"); m = torch.randn((N, max(K2, 4)), device="cuda")
    W = torch.randn((256, K1 + K2), device="cuda") * 0.1
    b = torch.randn(256, device="cuda")
    f = (lambda: gnnmp.dense(x, W[:, :K1], b, "relu", x2=m, W2=W[:, K1:])) if two else (lambda: gnnmp.dense(x, W, b, "relu"))
    res = {0: [], 64: []}
    ys = {}
    for rep in range(4):
        for kv in (0, 64):
            gnnmp.tune(19, kv)
            ys[kv] = f()
            res[kv].append(t(f))
    gnnmp.tune(19, 0)
    ref = (torch.cat([x, m], 1) if two else x).double() @ W.double().T + b.double()
    ref = torch.relu(ref)
    e0 = (ys[0].double() - ref).abs().max().item(); e64 = (ys[64].double() - ref).abs().max().item()
    print(f"N={N} K={K1}{'+' + str(K2) if two else ''} => 256: wreg {sorted(res[0])[1]*1e3:8.1f} us   split {sorted(res[64])[1]*1e3:8.1f} us   "
          f"|wreg-f64| {e0:.2e} |split-f64| {e64:.2e}  wreg==split {bool(torch.equal(ys[0], ys[64]))}", flush=True)
    del x, m, W, ref, ys
