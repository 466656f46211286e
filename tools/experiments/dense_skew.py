import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts)//2]
for (N, K, Dout) in [(2449029, 100, 100), (2449029, 100, 128), (169343, 128, 128)]:
    x = torch.randn((N, K), device="cuda"); W = torch.randn((Dout, K), device="cuda") * 0.1; b = torch.randn(Dout, device="cuda")
    res = []
    for sk in (0, 3, 16, 17, 19):
        gnnmp.tune(7, sk)
        res.append(f"knob{sk}: {t(lambda: gnnmp.dense(x, W, b, 'relu')):.3f}")
    gnnmp.tune(7, 0)
    print(f"N={N} K={K} Dout={Dout}: " + " | ".join(res))
