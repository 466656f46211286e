#!/usr/bin/env python
"""Experiment: SAGEConv(100 => 256) on the products shape with the aggregation of row range k + 1 (stream A) overlapping the
contraction of row range k (stream B) — the aggregation saturates the fabric, the contraction the matrix pipe and little else.
    python tools/experiments/sage_pipeline.py [parts ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gnnmp
from gnnmp import _lib as L, rowpart as RP, synth
from gnnmp.graph import Plan

parts_list = [int(v) for v in sys.argv[1:]] or [2, 4, 8]
N, D, Dout = synth.PRODUCTS["N"], synth.PRODUCTS["D"], 256
s, t = synth.products_like()
sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
g = gnnmp.GNNGraph(sd, td, num_nodes=N, _validated=True)
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
l = gnnmp.SAGEConv((D, Dout), "relu", seed=5)
lib = L.load()


def timeit(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


y0 = l(g, x)
print(f"whole graph, two kernels back to back: {timeit(lambda: l(g, x)):.3f} ms", flush=True)
W = l.weight
W1, W2 = W[:, :D], W[:, D:]
code = L.ACT_RELU
for C in parts_list:
    bounds = RP.partition_rows_by_edges(td, N, C)
    plans = []
    for lo, hi in bounds:
        sl, tl, _ = RP.local_edges(sd, td, lo, hi)
        plans.append(Plan(sl, tl, N, hi - lo, 1, False, validate=False))
    out = torch.empty((N, Dout), device="cuda")
    ms = [torch.empty((hi - lo, D), device="cuda") for lo, hi in bounds]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)      # the contraction's blocks go first when a CU frees up
    evs = [torch.cuda.Event() for _ in bounds]

    def step():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        for k, (lo, hi) in enumerate(bounds):
            L.check(lib.gnnmp_propagate_f32(plans[k].handle, L.COPY_XJ, L.MEAN, L.ptr(x), None, None, None, L.ptr(ms[k]), D, sa.cuda_stream))
            evs[k].record(sa)
            sb.wait_event(evs[k])
            xi = x[lo:hi]
            L.check(lib.gnnmp_dense_f32(L.ptr(xi), L.ptr(W1), D, W1.stride(0), L.ptr(ms[k]), L.ptr(W2), D, W2.stride(0), 0, L.ptr(l.bias), code,
                                        L.ptr(out[lo:hi]), hi - lo, Dout, sb.cuda_stream))
        cur.wait_stream(sa); cur.wait_stream(sb)
        return out

    y = step()
    torch.cuda.synchronize()
    err = float((y - y0).abs().max() / y0.abs().max())
    print(f"{C} row ranges, aggregation and contraction on two streams: {timeit(step):.3f} ms   max diff to whole graph {err:.1e}", flush=True)

    def serial():
        for k, (lo, hi) in enumerate(bounds):
            L.check(lib.gnnmp_propagate_f32(plans[k].handle, L.COPY_XJ, L.MEAN, L.ptr(x), None, None, None, L.ptr(ms[k]), D, L.stream_ptr()))
            L.check(lib.gnnmp_dense_f32(L.ptr(x[lo:hi]), L.ptr(W1), D, W1.stride(0), L.ptr(ms[k]), L.ptr(W2), D, W2.stride(0), 0, L.ptr(l.bias), code,
                                        L.ptr(out[lo:hi]), hi - lo, Dout, L.stream_ptr()))
    print(f"{C} row ranges, one stream: {timeit(serial):.3f} ms", flush=True)
    del plans
