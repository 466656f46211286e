#!/usr/bin/env python
"""Why does the one-pass attention kernel take 4.8 ms back to back but 5.2-5.3 ms inside the bench step?
Events around the attention launch ONLY, with different predecessors on the same stream (products shape, H = 8, C = 16):
    python tools/experiments/instep_probe.py [reps]
Every variant reports the median / min of the attention kernel's event time and, where it applies, of the GCN layer kernel's."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import _lib as L, synth
from gnnmp.layers import gcn_norm_cache

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lib = L.load()
N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
H, C = 8, 16
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
gcn = gnnmp.GCNConv((D, D), "relu", seed=11)
gat = gnnmp.GATConv((D, C), "relu", heads=H, seed=12)
cvec, c_slot, _ = gcn_norm_cache(g, True, None)
Wx = gnnmp.dense(x, gat.dense_x_weight)
Wx2 = Wx.clone()
a_hc = gat.a_hc
out = torch.empty_like(Wx)
scrub = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def k_gat(src=None):
    w = Wx if src is None else src
    L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(w), None, L.ptr(a_hc), 0.2, L.ptr(gat.bias), L.ACT_RELU, L.ptr(out), H, C, L.stream_ptr()))


def k_dense(dst=None):
    d = Wx if dst is None else dst
    W = gat.dense_x_weight
    L.check(lib.gnnmp_dense_f32(L.ptr(x), L.ptr(W), D, W.stride(0), None, None, 0, 0, 0, None, L.ACT_IDENTITY, L.ptr(d), N, H * C, L.stream_ptr()))


def k_gcn():
    return gnnmp.fused_conv(plan, L.SUM, x, gcn.weight, gcn.bias, "relu", ss_slot=c_slot, scale_dst=cvec)


def measure(name, pre, body=k_gat, post=None):
    for _ in range(2):
        pre(); body()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        pre()
        a.record(); body(); b.record()
        if post:
            post()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{name:70s} median {ts[len(ts) // 2]:.3f}  min {ts[0]:.3f}  max {ts[-1]:.3f} ms", flush=True)
    return ts[len(ts) // 2]


nop = lambda: None
print("--- attention kernel (gat_fused_rows_kernel) ---")
measure("A  alone, back to back", nop)
measure("B  after dense_split(x -> Wx) [the layer as it runs]", k_dense)
measure("C  after the GCN layer kernel", lambda: k_gcn())
measure("D  after GCN layer + dense (= the bench step)", lambda: (k_gcn(), k_dense()))
measure("E  after a 512 MiB memset (scrubs L2 + Infinity Cache)", lambda: scrub.zero_())
measure("H  after dense + 512 MiB memset", lambda: (k_dense(), scrub.zero_()))
measure("I  after dense writing ANOTHER buffer (gat reads a static Wx)", lambda: k_dense(Wx2))
measure("I2 after dense writing Wx, gat reading the static copy Wx2", k_dense, body=lambda: k_gat(Wx2))
lib.gnnmp_tune(17, -1)
measure("F  after the fp32-MFMA dense (knob 17 = -1)", k_dense)
lib.gnnmp_tune(17, 0)
measure("G  after a 20 ms idle gap", lambda: (torch.cuda.synchronize(), time.sleep(0.02)))
measure("G2 after a device-side spin of ~3 ms (torch.cuda._sleep)", lambda: torch.cuda._sleep(6_000_000))
measure("A' alone again", nop)
print("--- GCN layer kernel (fused_conv_kernel) ---")
measure("a  alone, back to back", nop, body=k_gcn)
measure("b  after the attention kernel", k_gat, body=k_gcn)
measure("c  after dense + attention (= the bench step)", lambda: (k_dense(), k_gat()), body=k_gcn)
measure("e  after a 512 MiB memset", lambda: scrub.zero_(), body=k_gcn)
print("--- dense_split (x -> Wx) ---")
measure("alone", nop, body=k_dense)
measure("after gcn", lambda: k_gcn(), body=k_dense)
# the kernel inside the LAYER / the STEP as bench.py runs them (the layer's own torch.empty allocations): events around the C call
real = lib.gnnmp_gat_conv_f32
pairs = []
class Spy:
    def __call__(self, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = real(*a); e1.record()
        pairs.append((e0, e1))
        return rc
lib.gnnmp_gat_conv_f32 = Spy()
for name, fn in (("L  inside gat(g, x), layers back to back", lambda: gat(g, x)), ("S  inside the bench step gcn(g, x); gat(g, x)", lambda: (gcn(g, x), gat(g, x)))):
    for _ in range(3):
        fn()
    pairs.clear()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in pairs)
    print(f"{name:70s} median {ts[len(ts) // 2]:.3f}  min {ts[0]:.3f}  max {ts[-1]:.3f} ms", flush=True)
lib.gnnmp_gat_conv_f32 = real
# whole step, wall clock, as bench.py times it
def step():
    gcn(g, x); gat(g, x)
for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"bench step (layers, wall clock over 20): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
