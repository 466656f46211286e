#!/usr/bin/env python
"""BASELINE config 4, SAGEConv(100 => 256, relu; mean) on the products shape: the two-kernel path (csr_rows_kernel + dense_wreg_kernel, with
and without placed buffers) against fused_cat_kernel's variants (knob 14: 16 = 12 waves a block, 8 = 8 waves), interleaved on ONE box.  python tools/experiments/sage_fused_ab.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, D = synth.PRODUCTS["N"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
sage = gnnmp.SAGEConv((D, 256), "relu", aggr="mean", seed=13)
g.plan(False)


def measure(it=20):
    for _ in range(3):
        sage(g, x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); sage(g, x); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], sum(ts) / len(ts)


ref = None
variants = [("two kernels, fresh buffers", -1, False), ("two kernels, placed m + out", -1, True), ("fused_cat 12 waves", 16, False),
            ("fused_cat 8 waves", 8, False)]
for r in range(rounds):
    for name, k14, placed in variants:
        gnnmp.tune(14, k14)
        sage.place_outputs = sage.persistent_out = placed
        med, avg = measure()
        y = sage(g, x)
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        print(f"round {r}: {name:32s} median {med:.3f} ms  mean {avg:.3f} ms   max |y - y_two_kernel| / max|y| = {err:.2e}", flush=True)
# ablations of the fused kernel (knob 13: 1 = no contraction, 2 = no gather, 4 = no stores) — wrong results, timings only
for k14 in (16, 8):
    for dbg, what in ((0, "whole kernel"), (1, "no contraction"), (2, "no gather"), (3, "neither (hand-out + stores of nothing)"), (4, "no stores")):
        gnnmp.tune(14, k14); gnnmp.tune(13, dbg)
        sage.place_outputs = sage.persistent_out = False
        med, avg = measure()
        print(f"ablation knob14={k14:2d} knob13={dbg}: {what:40s} median {med:.3f} ms", flush=True)
gnnmp.tune(13, 0)
gnnmp.tune(14, 0)
