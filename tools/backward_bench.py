#!/usr/bin/env python
"""Times the adjoint kernels on the products shape (D = 100): dxj via the transposed plan, dw via edge_dot, max gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
from gnnmp import synth, backward as bw

def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, tt = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(tt).cuda(), num_nodes=N, _validated=True)
x = torch.randn((N, D), device="cuda"); dy = torch.randn((N, D), device="cuda"); w = torch.rand(E, device="cuda")
y = gnnmp.propagate(gnnmp.copy_xj, g, "max", xj=x)
fwd = t(lambda: gnnmp.propagate(gnnmp.copy_xj, g, "+", xj=x))
dx = t(lambda: bw.propagate_grad_xj(g, "+", dy))
dxw = t(lambda: bw.propagate_grad_xj(g, "+", dy, w=w))
dw = t(lambda: bw.propagate_grad_w(g, dy, x))
# the COO-order comparator (gnnmp_edge_dot_f32 without a plan: what a caller gets who has no plan of the graph) only on request: it is not
# on any default path, and its 10-12 ms only cluttered the profile of the adjoints
if "--coo" in sys.argv:
    dwc = t(lambda: bw.propagate_grad_w(g, dy, x, coo_order=True))
    print(f"edge_dot: plan order {dw:.3f} ms, COO order {dwc:.3f} ms")
else:
    print(f"edge_dot: plan order {dw:.3f} ms")
dmax = t(lambda: bw.propagate_grad_xj(g, "max", dy, xj=x, y=y))
b1 = E * (4 * D + 4) + N * (4 * D + 8)
print(f"products D={D}: fwd(+) {fwd:.3f} ms | dxj(+) {dx:.3f} ms {b1/dx/1e6:.0f} GB/s | dxj(w) {dxw:.3f} ms | "
      f"dw edge_dot {dw:.3f} ms {E*(8*D+20)/dw/1e6:.0f} GB/s | dxj(max) {dmax:.3f} ms {E*(8*D+4)/dmax/1e6:.0f} GB/s")
from gnnmp.backward import dense_grad_w, dense_grad_x, act_grad, gcn_conv_ad
W = torch.randn((100, D), device="cuda") * 0.1
dz = torch.randn((N, 100), device="cuda")
tw = t(lambda: dense_grad_w(dz, x))
L_ = __import__("gnnmp")._lib.load()
L_.gnnmp_tune(10, -1); tw_old = t(lambda: dense_grad_w(dz, x)); L_.gnnmp_tune(10, 0)
print(f"dW+db 2.4M x 100 x 100: 16x16x4 kernel {tw:.3f} ms, round-1 32x32x2 kernel {tw_old:.3f} ms")
dz128 = torch.randn((N, 128), device="cuda")
tw128 = t(lambda: dense_grad_w(dz128, x))
print(f"dW+db 2.4M x 128 x 100: {tw128:.3f} ms ({2*N*128*D/tw128/1e9:.1f} TF)")
tx = t(lambda: dense_grad_x(dz, W))
ta = t(lambda: act_grad(dz, dz, "relu"))
print(f"dense adjoints 2.4M x 100 => 100: dW+db {tw:.3f} ms ({2*N*100*D/tw/1e9:.1f} TF) | dX {tx:.3f} ms | act_grad {ta:.3f} ms")
l = gnnmp.GCNConv((D, 100), "relu", seed=1); l.weight.requires_grad_(True); l.bias.requires_grad_(True)
xr = x.clone().requires_grad_(True)
def step():
    y = gcn_conv_ad(l, g, xr); y.backward(dy)
print(f"GCNConv forward+backward (autograd, HIP both ways): {t(step, 5):.3f} ms")
# GATConv(100 => 16, heads = 8, relu): fused forward with statistics + the two-pass adjoint
from gnnmp.backward import gat_conv_ad
from gnnmp import _lib as L
lg = gnnmp.GATConv((D, 16), "relu", heads=8, seed=2)
for prm in (lg.dense_x_weight, lg.a, lg.bias):
    prm.requires_grad_(True)
dyg = torch.randn((N, 128), device="cuda")
def gstep():
    y = gat_conv_ad(lg, g, xr); y.backward(dyg)
print(f"GATConv forward+backward (autograd, HIP both ways): {t(gstep, 5):.3f} ms")
# the attention adjoint alone
lib = L.load(); H, C = 8, 16
plan, plan_t = g.plan(True), bw.plan_transposed(g, True)
Wx = torch.randn((N, 128), device="cuda") * 0.3; a_hc = lg.a_hc.detach()
out = torch.empty((N, 128), device="cuda"); stats = torch.empty((N, H, 2), device="cuda")
tf = t(lambda: L.check(lib.gnnmp_gat_conv_stats_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, None, 0, L.ptr(out), L.ptr(stats), H, C, L.stream_ptr())))
line = torch.empty((N, H, 4), device="cuda"); dsd = torch.empty((N, H), device="cuda"); dss = torch.empty((N, H), device="cuda")
dWx = torch.empty((N, 128), device="cuda"); da = torch.empty((H, 2 * C), device="cuda")
tb = t(lambda: L.check(lib.gnnmp_gat_conv_grad_f32(plan.handle, plan_t.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, L.ptr(stats), L.ptr(dyg), L.ptr(line), L.ptr(dsd), L.ptr(dss), L.ptr(dWx), None, L.ptr(da), H, C, L.stream_ptr())))
Ep = E + N
alg = Ep * (512 + 4) + N * (3 * 512 + 64 + 32) + Ep * (512 + 128 + 4) + N * (2 * 512 + 64)
print(f"GAT attention: forward+stats {tf:.3f} ms | adjoint (2 edge passes + da) {tb:.3f} ms = {alg/tb/1e6:.0f} GB/s algorithmic")
# round 4: the training forward (out, stats, o+, P) and the one-edge-pass pullback (node kernel + source pass + da)
op = torch.empty((N, 128), device="cuda"); pp = torch.empty((N, H), device="cuda")
tf2 = t(lambda: L.check(lib.gnnmp_gat_conv_train_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, None, 0, L.ptr(out), L.ptr(stats), L.ptr(op), L.ptr(pp), H, C, L.stream_ptr())))
tb2 = t(lambda: L.check(lib.gnnmp_gat_conv_grad2_f32(plan.handle, plan_t.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, L.ptr(stats), L.ptr(out), None, L.ptr(op), L.ptr(pp), L.ptr(dyg), L.ptr(line), L.ptr(dsd), L.ptr(dss), L.ptr(dWx), None, L.ptr(da), H, C, L.stream_ptr())))
print(f"GAT attention, round 4: training forward (+ o+, P) {tf2:.3f} ms | adjoint (node kernel + 1 edge pass + da) {tb2:.3f} ms")

