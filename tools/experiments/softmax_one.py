#!/usr/bin/env python
"""softmax_edge_neighbors alone (gnnmp_edge_softmax_f32) on the products and arxiv shapes, H heads per edge: median of 11
launches, the one-pass narrow-row kernel (default) and the three-step kernels (knob 16 = -1) back to back on one box.
usage: softmax_one.py [H ...] [products|arxiv] [onepass]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

lib = L.load()
heads = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 1, 4]
only = [a for a in sys.argv[1:] if a in ("products", "arxiv")]
onepass_only = "onepass" in sys.argv[1:]


def med(fn, n=11):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


shapes = (("products", synth.products_like, synth.PRODUCTS["N"]), ("arxiv", synth.arxiv_like, synth.ARXIV["N"]))
for name, gen, N in shapes:
    if only and name not in only:
        continue
    s, t = gen()
    g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
    E = g.num_edges
    for H in heads:
        e = torch.randn((E, H), device="cuda")
        res = {}
        outs = {}
        for label, k in (("one-pass", 0),) if onepass_only else (("one-pass", 0), ("three-step", -1)):
            lib.gnnmp_tune(16, k)
            res[label] = med(lambda: gnnmp.softmax_edge_neighbors(g, e))
            outs[label] = gnnmp.softmax_edge_neighbors(g, e)
        lib.gnnmp_tune(16, 0)
        alg = E * 8 * H + E * 4
        if onepass_only:
            print(f"{name} H={H}: one-pass {res['one-pass']:.3f} ms ({alg / res['one-pass'] / 1e6:.0f} GB/s alg)", flush=True)
        else:
            same = torch.equal(outs["one-pass"], outs["three-step"])
            print(f"{name} H={H}: one-pass {res['one-pass']:.3f} ms ({alg / res['one-pass'] / 1e6:.0f} GB/s alg)  "
                  f"three-step {res['three-step']:.3f} ms  bit-identical {same}", flush=True)
        del e, outs
