#!/usr/bin/env python
"""Experiment: SAGEConv(100 => 256, mean) on the products shape with the aggregation of row range k + 1 and the
contraction of row range k on DISJOINT compute units — two streams made with hipExtStreamCreateWithCUMask (tools/ubench/cumask.hip).
The plain two-stream pipeline (tools/experiments/sage_pipeline.py) gave no overlap: the row kernel owned every wave slot.
    python tools/experiments/sage_cumask.py [parts,dense_cus ...]      e.g. 8,48 8,64 16,48"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import gnnmp
from gnnmp import _lib as L, rowpart as RP, synth
from gnnmp.graph import Plan

cm = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libcumask.so"))
cm.cm_stream_create.restype = ctypes.c_void_p
cm.cm_stream_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
cm.cm_event_create.restype = ctypes.c_void_p
for f in (cm.cm_event_record, cm.cm_stream_wait):
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
cm.cm_stream_sync.argtypes = [ctypes.c_void_p]
cm.cm_stream_destroy.argtypes = [ctypes.c_void_p]

cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(8, 48), (8, 64), (16, 48), (16, 32), (4, 64), (8, 96)]
N, D, Dout = synth.PRODUCTS["N"], synth.PRODUCTS["D"], 256
CUS = torch.cuda.get_device_properties(0).multi_processor_count
s, t = synth.products_like()
sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
g = gnnmp.GNNGraph(sd, td, num_nodes=N, _validated=True)
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
l = gnnmp.SAGEConv((D, Dout), "relu", seed=5)
lib = L.load()
W = l.weight
W1, W2 = W[:, :D], W[:, D:]
code = L.ACT_RELU


def timeit(fn, sync, it=10):
    fn(); fn(); sync()
    ts = []
    for _ in range(it):
        sync(); t0 = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


y0 = l(g, x)
med, mn = timeit(lambda: l(g, x), torch.cuda.synchronize)
print(f"CUs {CUS}; whole graph, two kernels back to back: median {med:.3f} min {mn:.3f} ms", flush=True)
plans_by_parts = {}
for parts, dcus in cfgs:
    if parts not in plans_by_parts:
        bounds = RP.partition_rows_by_edges(td, N, parts)
        pl = []
        for lo, hi in bounds:
            sl, tl, _ = RP.local_edges(sd, td, lo, hi)
            pl.append(Plan(sl, tl, N, hi - lo, 1, False, validate=False))
        plans_by_parts[parts] = (bounds, pl)
    bounds, plans = plans_by_parts[parts]
    out = torch.empty((N, Dout), device="cuda")
    ms = [torch.empty((hi - lo, D), device="cuda") for lo, hi in bounds]
    torch.cuda.synchronize()
    for mode in ("masked", "masked_agg_all", "unmasked"):
        if mode == "masked":
            sa, sb = cm.cm_stream_create(dcus, CUS, CUS), cm.cm_stream_create(0, dcus, CUS)
        elif mode == "masked_agg_all":      # the contraction confined, the aggregation free to use every CU
            sa, sb = cm.cm_stream_create(0, CUS, CUS), cm.cm_stream_create(0, dcus, CUS)
        else:
            sa, sb = cm.cm_stream_create(0, CUS, CUS), cm.cm_stream_create(0, CUS, CUS)
        assert sa and sb
        evs = [cm.cm_event_create() for _ in bounds]

        def step():
            for k, (lo, hi) in enumerate(bounds):
                L.check(lib.gnnmp_propagate_f32(plans[k].handle, L.COPY_XJ, L.MEAN, L.ptr(x), None, None, None, L.ptr(ms[k]), D, ctypes.c_void_p(sa)))
                cm.cm_event_record(evs[k], sa)
                cm.cm_stream_wait(sb, evs[k])
                xi = x[lo:hi]
                L.check(lib.gnnmp_dense_f32(L.ptr(xi), L.ptr(W1), D, W1.stride(0), L.ptr(ms[k]), L.ptr(W2), D, W2.stride(0), 0, L.ptr(l.bias), code,
                                            L.ptr(out[lo:hi]), hi - lo, Dout, ctypes.c_void_p(sb)))

        def sync():
            cm.cm_stream_sync(sa); cm.cm_stream_sync(sb)

        step(); sync()
        err = float((out - y0).abs().max() / y0.abs().max())
        med, mn = timeit(step, sync)
        print(f"parts {parts:2d}  dense CUs {dcus:3d}  {mode:15s}: median {med:.3f} min {mn:.3f} ms   max diff {err:.1e}", flush=True)
        # the two halves alone on their streams (what each costs under its mask)
        def agg_only():
            for k in range(len(bounds)):
                L.check(lib.gnnmp_propagate_f32(plans[k].handle, L.COPY_XJ, L.MEAN, L.ptr(x), None, None, None, L.ptr(ms[k]), D, ctypes.c_void_p(sa)))
        def dense_only():
            for k, (lo, hi) in enumerate(bounds):
                L.check(lib.gnnmp_dense_f32(L.ptr(x[lo:hi]), L.ptr(W1), D, W1.stride(0), L.ptr(ms[k]), L.ptr(W2), D, W2.stride(0), 0, L.ptr(l.bias), code,
                                            L.ptr(out[lo:hi]), hi - lo, Dout, ctypes.c_void_p(sb)))
        if mode != "unmasked":
            ma, _ = timeit(agg_only, sync)
            md, _ = timeit(dense_only, sync)
            print(f"          alone under the masks: aggregation {ma:.3f} ms, contraction {md:.3f} ms", flush=True)
        cm.cm_stream_destroy(sa); cm.cm_stream_destroy(sb)
    del out, ms
