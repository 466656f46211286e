#!/usr/bin/env python
"""graph_chain2_kernel launch variants (knob 19, read at gnnmp_chain_jobs_create and at launch): bit 3 = 32-row jobs (one MFMA tile a
wave; low bits 0: 16 waves a block, 1: 12, 2: 8), else 64-row jobs (low bits 0: 8 waves, 1: 4, 2: 6); bit 2 = scheduling barriers.
usage: chain_variants.py [G] [nmin] [nmax]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gnnmp
from gnnmp import synth


def t(fn, it=50):
    fn(); fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nmin = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 40
members = synth.batched_graphs(G=G, nmin=nmin, nmax=nmax)
rng = np.random.default_rng(4)
xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
y0 = None
variants = [int(v) for v in os.environ.get('VARIANTS', '3,0,1,2').split(',')]
for kv in variants:
    gnnmp.tune(19, kv)
    g = gnnmp.batch_arrays(members, xs)          # (a fresh graph: the jobs are built under this knob value)
    f = lambda: model(g, g.x)
    y = f()
    if y0 is None:
        y0 = y
    cj = g._cache.get("chain_jobs"); info = (cj.njobs, cj.fill) if cj is not None else None
    print(f"chain2 knob19={kv}: {t(f)*1e3:7.1f} us  max diff {float((y - y0).abs().max()):.1e}  jobs {info}", flush=True)
gnnmp.tune(19, 0)
