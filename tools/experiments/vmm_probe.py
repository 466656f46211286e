#!/usr/bin/env python
"""Placement classes through the HIP virtual-memory API: physical chunks of CH GiB created one by one (hipMemCreate), each mapped at its own
address; every chunk classified by timing the attention kernel (products shape) with a reference chunk as the source and the chunk as the
output (slow = same class as the reference).  Then: does a SMALL probe (a propagate on a synthetic regular graph) tell the classes apart?
    python tools/experiments/vmm_probe.py [chunks] [chunk_GiB]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch, gnnmp
from gnnmp import _lib as L, synth
from gnnmp.graph import Plan

NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 100
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
GiB = 1 << 30
vm = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libvmm.so"))
u64 = ctypes.c_uint64
vm.vmm_granularity.restype = u64
vm.vmm_chunk_create.restype = u64; vm.vmm_chunk_create.argtypes = [u64, ctypes.c_int]
vm.vmm_reserve.restype = u64; vm.vmm_reserve.argtypes = [u64, u64]
vm.vmm_map.argtypes = [u64, u64, u64, ctypes.c_int]
torch.cuda.init(); torch.zeros(1, device="cuda")
print("granularity", vm.vmm_granularity(0))

lib = L.load()
N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
s, t = synth.products_like()
g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N, _validated=True)
plan = g.plan(True)
H, C = 8, 16
HC = H * C
x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
gat = gnnmp.GATConv((D, C), "relu", heads=H, seed=12)
Wx0 = gnnmp.dense(x, gat.dense_x_weight)
a_hc = gat.a_hc
nbytes = N * HC * 4
assert nbytes <= CH * GiB

base = vm.vmm_reserve(NCH * CH * GiB, 2 << 20)
assert base
handles = []
for i in range(NCH):
    h = vm.vmm_chunk_create(CH * GiB, 0)
    if not h:
        print("hipMemCreate stopped at chunk", i); NCH = i; break
    assert vm.vmm_map(base + i * CH * GiB, CH * GiB, h, 0) == 0
    handles.append(h)
print(f"{NCH} chunks of {CH} GiB mapped at {base:#x}; free now {torch.cuda.mem_get_info()[0] / GiB:.1f} GiB", flush=True)


class Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}


def chunk_tensor(i, rows=N, cols=HC):
    return torch.as_tensor(Raw(base + i * CH * GiB, (rows, cols)), device="cuda")


def timed(f, reps=4):
    f(); f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def gatk(src, dst):
    return lambda: L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(src), None, L.ptr(a_hc), 0.2, L.ptr(gat.bias), L.ACT_RELU, L.ptr(dst), H, C, L.stream_ptr()))


cls = [-1] * NCH
ref = 0
ncls = 0
times = {}
while -1 in cls and ncls < 6:
    src = chunk_tensor(ref); src.copy_(Wx0)
    cls[ref] = ncls
    row = []
    for j in range(NCH):
        if cls[j] != -1 and j != ref:
            row.append(None); continue
        if j == ref:
            row.append(None); continue
        tm = timed(gatk(src, chunk_tensor(j)))
        row.append(tm)
    vals = [v for v in row if v is not None]
    if not vals:
        break
    lo, hi = min(vals), max(vals)
    thr = (lo + hi) / 2 if hi > lo * 1.03 else hi + 1
    for j, v in enumerate(row):
        if v is not None and v > thr:
            cls[j] = ncls
    print(f"reference chunk {ref}: fastest {lo:.2f} slowest {hi:.2f} ms; class {ncls} has {cls.count(ncls)} chunks", flush=True)
    times[ncls] = row
    ncls += 1
    rest = [j for j in range(NCH) if cls[j] == -1]
    if not rest:
        break
    ref = rest[0]
print("class of every chunk in creation order: " + "".join("ABCDEF?"[c] for c in cls))
# a small probe: propagate(copy_xj, +) on a regular random graph, source in class A's reference chunk, output in a chunk of every class
rng = np.random.default_rng(0)
NS_FULL = CH * GiB // 512
for n_rows, deg, n_src, zero in ((1 << 16, 26, N, False), (1 << 18, 26, N, False), (1 << 18, 26, NS_FULL, False), (1 << 18, 26, NS_FULL, True)):
    dst = np.repeat(np.arange(n_rows, dtype=np.int64), deg) + 1
    srcn = rng.integers(0, n_src, n_rows * deg).astype(np.int64) + 1
    pp = Plan(torch.from_numpy(srcn).cuda(), torch.from_numpy(dst).cuda(), n_src, n_rows, 1, False, validate=False)
    xs = chunk_tensor(0, n_src, HC)
    if zero:
        xs.zero_()
    line = []
    for c in range(ncls):
        js = [j for j in range(1, NCH) if cls[j] == c][:3]
        for j in js:
            o = chunk_tensor(j, n_rows, HC)
            f = lambda: L.check(lib.gnnmp_propagate_f32(pp.handle, L.COPY_XJ, L.SUM, L.ptr(xs), None, None, None, L.ptr(o), HC, L.stream_ptr()))
            line.append(f"{'ABCDEF'[c]}{j}:{timed(f, 9) * 1e3:.1f}")
    print(f"propagate probe {n_rows} rows x {deg} random sources of {n_src} (zeroed {zero}) (source = chunk 0, class A), output chunk -> us: " + " ".join(line), flush=True)
