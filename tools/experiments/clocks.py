#!/usr/bin/env python
"""Engine clock, socket power and time per launch under SUSTAINED back-to-back launches (2.5 s) of one kernel of the step:
    python tools/experiments/clocks.py gat | gcn | dense100 | dense128 | mfma
rocm-smi is sampled once, one second into the run.  (Event-timed single launches after an idle gap run at other clocks.)"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth

what = sys.argv[1:] or ["gat", "gcn", "dense100", "dense128"]
lib = L.load()
N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
H, C = 8, 16
x = torch.randn((N, D), device="cuda")
Wx = torch.randn((N, H * C), device="cuda") * 0.3
a = torch.randn((H, 2 * C), device="cuda") * 0.3
b = torch.randn(H * C, device="cuda") * 0.1
out = torch.empty_like(Wx)
gcn = gnnmp.GCNConv((D, D), "relu", seed=1)
W128 = torch.randn((128, D), device="cuda") * 0.1


def smi():
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    keep = [l.split(":", 1)[1].strip() if ":" in l else l.strip() for l in o.splitlines()
            if any(k in l for k in ("sclk", "mclk", "fclk", "Power (W)"))]
    return " | ".join(keep)


runs = {
    "gat": lambda: L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, L.ptr(b), L.ACT_RELU, L.ptr(out), H, C, L.stream_ptr())),
    "gcn": lambda: gcn(g, x),
    "dense100": lambda: gnnmp.dense(x, gcn.weight, gcn.bias, "relu"),
    "dense128": lambda: gnnmp.dense(x, W128),
}
for name in what:
    run = runs[name]
    run(); torch.cuda.synchronize()
    res = {}
    th = threading.Thread(target=lambda: (time.sleep(1.0), res.__setitem__("smi", smi())))
    th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(20):
            run()
        torch.cuda.synchronize(); n += 20
    dt = (time.perf_counter() - t0) / n * 1e3
    th.join()
    print(f"{name}: {dt:.3f} ms/launch sustained over {n} launches   [{res.get('smi')}]", flush=True)
