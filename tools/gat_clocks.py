import os, sys, subprocess, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth
lib = L.load()
N, E = synth.PRODUCTS["N"], synth.PRODUCTS["E"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
H, C = 8, 16
Wx = torch.randn((N, H * C), device="cuda") * 0.3
a = torch.randn((H, 2 * C), device="cuda") * 0.3
b = torch.randn(H * C, device="cuda") * 0.1
out = torch.empty_like(Wx)
def run():
    L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(Wx), None, L.ptr(a), 0.2, L.ptr(b), L.ACT_RELU, L.ptr(out), H, C, L.stream_ptr()))
def smi():
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in o.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "memory)"))]
    return " | ".join(keep)
for ro in (0, 1, 0, 1):
    gnnmp.tune(15, ro)
    run(); torch.cuda.synchronize()
    res = {}
    def sampler():
        time.sleep(1.0)
        res["smi"] = smi()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(20): run()
        torch.cuda.synchronize(); n += 20
    dt = (time.perf_counter() - t0) / n * 1e3
    th.join()
    print(f"row_order={ro}: {dt:.3f} ms/launch over {n} launches   {res.get('smi')}", flush=True)
