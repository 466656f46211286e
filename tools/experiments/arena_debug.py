#!/usr/bin/env python
"""The arena's classification, chunk by chunk (GNNMP_ARENA_DEBUG=1), and the classes of a few torch allocations.
    GNNMP_ARENA_DEBUG=1 python tools/experiments/arena_debug.py [gib_per_class]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import placement
os.environ.setdefault("GNNMP_ARENA_DEBUG", "1")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8
ar = placement.Arena(gib_per_class=gib)
print(ar.info())
for i in range(6):
    x = torch.randn((2449029, 100), device="cuda")
    print("torch buffer", i, hex(x.data_ptr()), "class", ar.class_of(x))
