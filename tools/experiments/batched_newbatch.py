#!/usr/bin/env python
"""Config 5 as the reference's loop runs it — a NEW batch every step (examples/graph_classification_tudataset.jl:70-71, 97-104) — from the
device-resident dataset: where the time of one step goes.  Host sections with perf_counter (no synchronisation inside), the whole loop
pipelined and synchronised.  Run under tools/profile_cmd.sh for the kernels' own durations.
    python tools/experiments/batched_newbatch.py [G] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gnnmp
from gnnmp import synth, layers

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
members = synth.batched_graphs(G=G)
rng = np.random.default_rng(4)
xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
ds = gnnmp.GraphDataset.from_members(members, xs)
model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                       gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
loader = gnnmp.DataLoader(ds, batchsize=G, shuffle=True, seed=7)
for _ in range(5):
    for g in loader:
        model(g, g.x)
torch.cuda.synchronize()


def wall(fn, n=iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def host(fn, n=iters):      # host time only: the device is drained first, then only the enqueue side is clocked
    tot = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); tot += time.perf_counter() - t0
    torch.cuda.synchronize()
    return tot / n * 1e6


perm_rng = np.random.default_rng(1)
state = {}


def s_perm():
    state["perm"] = perm_rng.permutation(G)


def s_upload():
    state["dev"] = torch.from_numpy(state["perm"] + 1).to("cuda")


def s_batch():
    state["g"] = ds.batch(state["perm"], state["dev"])


def s_jobs():
    g = state["g"]
    state["j"] = layers.ChainJobs(g._cache["node_ptr"], g.num_graphs, g._cache["member_stats"])
    g._cache["chain_jobs"] = state["j"]


def s_model():
    g = state["g"]
    state["y"] = model(g, g.x)


s_perm(); s_upload(); s_batch(); s_jobs(); s_model()
print(f"G = {G}: host microseconds per section (device idle at entry, enqueue side only)")
for name, fn in (("numpy permutation", s_perm), ("upload of the permutation", s_upload), ("GraphDataset.batch (select + gather enqueue)", s_batch),
                 ("ChainJobs pack enqueue", s_jobs), ("model(g, x) enqueue", s_model)):
    print(f"  {name:48s} {host(fn):8.1f} us")
print("wall microseconds per section, synchronised after each call (host + device latency)")
for name, fn in (("upload of the permutation", s_upload), ("GraphDataset.batch", s_batch), ("ChainJobs pack", s_jobs), ("model(g, x) (jobs ready)", s_model)):
    print(f"  {name:48s} {host(lambda: (fn(), torch.cuda.synchronize())):8.1f} us")


def full():
    for g in loader:
        model(g, g.x)


print(f"whole loop, pipelined (no synchronisation): {wall(full):8.1f} us per step")
print(f"whole loop, synchronised every step:        {host(lambda: (full(), torch.cuda.synchronize())):8.1f} us per step")
g = state["g"]
print(f"step alone on one batch object:              {wall(lambda: model(g, g.x)):8.1f} us")
