#!/usr/bin/env python
"""dense_split_kernel A/B runs on one box (knob 19): bit 4 (16) = one column tile after the other instead of side by side (SAGEConv's
256 columns), bit 5 (32) = stores straight from the accumulator layout instead of through the per-wave LDS stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import gnnmp


def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for (N, K, Dout, two) in [(2449029, 100, 128, False), (2449029, 100, 256, True), (169343, 128, 128, False), (245246, 16, 128, True),
                          (245246, 128, 128, True), (100000, 52, 36, False), (100000, 24, 200, True), (33, 8, 64, False)]:
    x = torch.randn((N, K), device="cuda"); m = torch.randn((N, K), device="cuda")
    W = torch.randn((Dout, 2 * K if two else K), device="cuda") * 0.1
    b = torch.randn(Dout, device="cuda")
    f = (lambda: gnnmp.dense(x, W[:, :K], b, "relu", x2=m, W2=W[:, K:])) if two else (lambda: gnnmp.dense(x, W, b, "relu"))
    y0 = f()
    row = []
    res = {0: [], 32: [], 16: []}
    same = True
    for rep in range(4):                      # interleaved: clocks drift by 10 % over the first seconds of a run
        for kv in (0, 32, 16):
            gnnmp.tune(19, kv)
            same = same and bool(torch.equal(f(), y0))
            res[kv].append(t(f, 10))
    for kv in (0, 32, 16):
        row.append(f"knob19={kv}: {sorted(res[kv])[1]*1e3:8.1f} us")
    row.append(f"equal={same}")
    gnnmp.tune(19, 0)
    print(f"N={N} K={K}{'+' + str(K) if two else ''} Dout={Dout}: " + "   ".join(row), flush=True)
