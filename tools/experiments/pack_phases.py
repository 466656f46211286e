#!/usr/bin/env python
"""Phase stamps of chain_pack_kernel (100 MHz wall clock, thread 0): where the per-batch job packing spends its time.
    python tools/experiments/pack_phases.py [G]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, gnnmp
from gnnmp import layers

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(3)
sizes = rng.integers(20, 41, G).astype(np.int64)
seg = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)])).cuda()
names = ["zero tables", "count pass (loads + LDS atomics)", "prefix of counts", "sort pass (ranks + scattered stores)", "records (one wave, tables in lanes)",
         "class lists", "state to memory"]
acc = np.zeros(6)
for it in range(20):
    j = layers.ChainJobs(seg, G, (int(sizes.sum()), int(sizes.max()), False))
    _, hdr = j.export()
    st = hdr.cpu().numpy()[16:30].view(np.int64)
    if it >= 5:
        acc += np.diff(st) / 100.0
    del j
for n, v in zip(names[1:], acc / 15):
    print(f"  {n:44s} {v:7.2f} us")
print(f"  total inside the kernel                      {acc.sum() / 15:7.2f} us")
