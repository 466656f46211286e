import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch, gnnmp
N, K, Dout = 2449029, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 100
x = torch.randn((N, K), device="cuda"); W = torch.randn((Dout, K), device="cuda") * 0.1; b = torch.randn(Dout, device="cuda")
for _ in range(3):
    gnnmp.dense(x, W, b, "relu")
torch.cuda.synchronize()
