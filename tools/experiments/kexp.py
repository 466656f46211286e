#!/usr/bin/env python
"""Kernel experiments on the GPU box: times gnnmp_propagate_f32 / gnnmp_gat_aggregate_f32 under different tuning knobs
and graph structures (power-law vs regular vs sequential sources) to locate the limiter.  Prints a table to stdout.
Perf-engineering tool, not part of the product or the tests."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import gnnmp  # noqa: E402
from gnnmp import _lib as L  # noqa: E402
from gnnmp import synth  # noqa: E402

KNOBS = {"vec": 0, "log2g": 1, "unroll": 2, "xcd": 3, "long": 4, "waves": 5}
DEFAULTS = {"vec": 0, "log2g": -1, "unroll": 0, "xcd": 1, "long": 0, "waves": 0}


def set_knobs(**kw):
    for k, v in DEFAULTS.items():
        gnnmp.tune(KNOBS[k], kw.get(k, v))


def time_fn(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def make_graph(kind, N, E, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "products":
        return synth.products_like(N=N, E=E)
    if kind == "arxiv":
        return synth.arxiv_like(N=N, E=E)
    deg = E // N
    t = np.repeat(np.arange(N, dtype=np.int64), deg)
    if kind == "regular_random":
        s = rng.integers(0, N, size=N * deg, dtype=np.int64)
    elif kind == "regular_seq":          # sources consecutive: nearly streaming reads
        s = (t * deg + np.tile(np.arange(deg, dtype=np.int64), N)) % N
    elif kind == "regular_local":        # sources within +-4096 of the destination
        s = (t + rng.integers(-4096, 4096, size=N * deg)) % N
    else:
        raise ValueError(kind)
    p = rng.permutation(len(s))
    return s[p] + 1, t[p] + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=synth.PRODUCTS["N"])
    ap.add_argument("--E", type=int, default=synth.PRODUCTS["E"])
    ap.add_argument("--D", type=int, default=100)
    ap.add_argument("--graphs", default="products,regular_random,regular_seq,regular_local")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    N, E, D = args.N, args.E, args.D
    lib = L.load()
    x = torch.randn((N, D), device="cuda")
    out = torch.empty((N, D), device="cuda")
    H, C = 8, 16
    Wx = torch.randn((N, H * C), device="cuda")
    sd = torch.randn((N, H), device="cuda")
    ss = torch.randn((N, H), device="cuda")
    og = torch.empty((N, H * C), device="cuda")
    cv = torch.rand(N, device="cuda") + 0.5
    # streaming-copy ceiling for reference
    big = torch.empty(256 * 1024 * 1024, device="cuda")
    big2 = torch.empty_like(big)
    tc = time_fn(lambda: big2.copy_(big))
    print(f"copy 1 GiB -> 1 GiB: {tc:.3f} ms = {2 * big.numel() * 4 / tc / 1e6:.0f} GB/s")
    del big, big2
    for kind in args.graphs.split(","):
        s, t = make_graph(kind, N, E)
        Ecur = len(s)
        set_knobs()
        g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
        plan = g.plan(False)
        bytes_p = Ecur * (4 * D + 4) + 8 * (N + 1) + 4 * N * D
        bytes_g = Ecur * (4 * H * C + 4) + N * (8 * H * C + 8)

        cslot = torch.empty(plan.n_total, device="cuda")
        L.check(lib.gnnmp_plan_slot_gather_f32(plan.handle, 0, L.ptr(cv), L.ptr(cslot), L.stream_ptr()))
        a_hc = torch.randn((H, 2 * C), device="cuda") * 0.3

        def prop(scaled=False):
            if scaled:
                L.check(lib.gnnmp_propagate_slots_f32(plan.handle, 0, L.ptr(x), None, L.ptr(cslot), L.ptr(cv),
                                                      L.ptr(out), D, L.stream_ptr()))
            else:
                L.check(lib.gnnmp_propagate_f32(plan.handle, 0, 0, L.ptr(x), None, None, None, L.ptr(out), D,
                                                L.stream_ptr()))

        def gat3():
            L.check(lib.gnnmp_gat_aggregate_f32(plan.handle, L.ptr(Wx), L.ptr(sd), L.ptr(ss), 0.2, None, 0, L.ptr(og),
                                                None, H, C, L.stream_ptr()))

        def gat():
            L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, None, 0, L.ptr(og), H, C,
                                           L.stream_ptr()))

        print(f"== {kind}: N={N} E={Ecur} D={D} maxdeg={plan.max_degree} long={plan.n_long}")
        variants = [dict()]
        if not args.quick:
            variants += [dict(unroll=2), dict(unroll=8), dict(waves=1), dict(waves=2), dict(xcd=0),
                         dict(xcd=0, unroll=8), dict(xcd=0, waves=1), dict(xcd=0, waves=2), dict(xcd=0, unroll=8, waves=1),
                         dict(xcd=0, unroll=8, waves=2), dict(vec=2, log2g=6)]
        for v in variants:
            set_knobs(**v)
            tp = time_fn(prop)
            tps = time_fn(lambda: prop(True))
            tg = time_fn(gat)
            tg3 = time_fn(gat3)
            print(f"  {str(v):48s} prop {tp:7.3f} ms {bytes_p / tp / 1e6:6.0f} GB/s | scaled {tps:7.3f} ms | "
                  f"gat {tg:7.3f} ms {bytes_g / tg / 1e6:6.0f} GB/s | gat3pass {tg3:7.3f} ms")
        set_knobs()
        del g, plan


if __name__ == "__main__":
    main()
