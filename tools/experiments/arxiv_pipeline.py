#!/usr/bin/env python
"""Does the arxiv-shaped GCN layer's contraction hide behind its gather when the destination rows are cut in K ranges and the two
kernels run on two streams (gather of range k + 1 next to the contraction of range k)?  Sub-plans per row range (bipartite plans:
n_src = N, n_dst = rows of the range) through the existing C ABI; events between the streams.  Prints sequential vs pipelined times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gnnmp
from gnnmp import _lib as L, synth

lib = L.load()
N, D = synth.ARXIV["N"], synth.ARXIV["D"]
s, t = synth.arxiv_like()
s = np.concatenate([s, np.arange(1, N + 1)]); t = np.concatenate([t, np.arange(1, N + 1)])      # self loops as edges
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
W = torch.randn((D, D), device="cuda") * 0.1
b = torch.zeros(D, device="cuda")
c = torch.rand(N, device="cuda") + 0.5
sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
full = gnnmp.Plan(sd, td, N, N, 1, False)


def prop(plan, c_dst, out, stream):
    L.check(lib.gnnmp_propagate_f32(plan.handle, L.COPY_XJ, L.SUM, L.ptr(x), None, L.ptr(c), L.ptr(c_dst), L.ptr(out), D, stream))


def dense(xin, out, stream):
    L.check(lib.gnnmp_dense_f32(L.ptr(xin), L.ptr(W), D, W.stride(0), None, None, 0, 0, 0, L.ptr(b), L.ACT_RELU, L.ptr(out), xin.shape[0], D, stream))


def timeit(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, bb in ev:
        a.record(); fn(); bb.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(bb) for a, bb in ev)
    return ts[len(ts) // 2] * 1e3


y = torch.empty((N, D), device="cuda"); z = torch.empty((N, D), device="cuda")
s0 = torch.cuda.current_stream()
sp = L.stream_ptr()


def sequential():
    prop(full, c, y, sp)
    dense(y, z, sp)


t_seq = timeit(sequential)
t_prop = timeit(lambda: prop(full, c, y, sp))
t_dense = timeit(lambda: dense(y, z, sp))
z_ref = z.clone()
print(f"sequential {t_seq:7.1f} us   (gather alone {t_prop:.1f}, contraction alone {t_dense:.1f})", flush=True)

side = torch.cuda.Stream()
for K in (2, 3, 4, 6):
    bounds = [int(round(N * k / K)) for k in range(K + 1)]
    # balance by edges rather than rows: equal slot counts per range
    order = np.argsort(t, kind="stable")
    tt = t[order]
    cuts = [0] + [int(tt[min(len(tt) - 1, (len(tt) * k) // K)]) for k in range(1, K)] + [N]
    bounds = cuts
    plans, cs = [], []
    for k in range(K):
        lo, hi = bounds[k], bounds[k + 1]
        m = (t > lo) & (t <= hi)
        plans.append(gnnmp.Plan(torch.from_numpy(s[m]).cuda(), torch.from_numpy(t[m] - lo).cuda(), N, hi - lo, 1, False))
        cs.append(c[lo:hi].contiguous())
    evs = [torch.cuda.Event() for _ in range(K)]
    done = torch.cuda.Event()

    def pipelined():
        for k in range(K):
            lo, hi = bounds[k], bounds[k + 1]
            prop(plans[k], cs[k], y[lo:hi], sp)
            evs[k].record(s0)
            side.wait_event(evs[k])
            with torch.cuda.stream(side):
                dense(y[lo:hi], z[lo:hi], L.stream_ptr())
        done.record(side)
        s0.wait_event(done)

    def chunks_one_stream():
        for k in range(K):
            lo, hi = bounds[k], bounds[k + 1]
            prop(plans[k], cs[k], y[lo:hi], sp)
            dense(y[lo:hi], z[lo:hi], sp)

    z.zero_()
    t_pipe = timeit(pipelined)
    ok = bool(torch.equal(z, z_ref))
    t_one = timeit(chunks_one_stream)
    print(f"K={K}: two streams {t_pipe:7.1f} us   same chunks on one stream {t_one:7.1f} us   identical={ok}", flush=True)
