"""Generates tests/golden/config1_cora_v1.npz — BASELINE.json config 1: the reference's Cora node-classification model
`GNNChain(GCNConv(1433 => 64, relu), GCNConv(64 => 64, relu), Dense(64 => 7))`
(GraphNeuralNetworks/test/examples/node_classification_cora.jl:18-24,51-54: nhidden = 64, seed = 17) on a Cora-shaped
synthetic graph (MLDatasets is unreachable: N = 2 708, E = 10 556 bidirected, 1 433 sparse row-normalised features,
GNNGraphs/src/mldatasets.jl:14-21).

Expected outputs come from the CPU oracle along the path the reference takes with CPU arrays: every GCNConv propagate is
the SpMM fast path `xj * adjacency_matrix(g)` (GNNlib/src/msgpass.jl:215-238, CSC order, duplicate edges pre-summed);
layer 1 has Dout < Din so the weight is applied first (GNNlib/src/layers/conv.jl:36-40).  The fixture is DATA (inputs,
weights, expected activations); nothing of the reference's source is stored.

    python tests/golden/make_golden_configs.py      # deterministic (seeded)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graphneuralnetworks.jl_amd", "gnnmp"))
from oracle import oracle as orc  # noqa: E402
import synth  # noqa: E402  (numpy only; imported as a plain module so that no GPU library is touched)

NHIDDEN, SEED = 64, 17


def glorot(rng, rows, cols):
    """Flux.glorot_uniform(rows, cols): U(-s, s), s = sqrt(6 / (rows + cols))"""
    s = np.sqrt(6.0 / (rows + cols))
    return ((rng.random((rows, cols)) * 2 - 1) * s).astype(np.float32)


def cora_model(blas=False):
    s, t, x = synth.cora_like(seed=SEED)
    n, nin, nout = synth.CORA["N"], synth.CORA["D"], synth.CORA["C"]
    rng = np.random.default_rng(SEED)
    W1, W2, Wd = glorot(rng, NHIDDEN, nin), glorot(rng, NHIDDEN, NHIDDEN), glorot(rng, nout, NHIDDEN)
    # Flux initialises biases to zero; small non-zero ones exercise the `.+ bias` of every layer
    b1, b2, bd = (rng.standard_normal(k).astype(np.float32) * 0.05 for k in (NHIDDEN, NHIDDEN, nout))
    c = dict(s=s, t=t, n=np.int64(n), x=x, W1=W1, b1=b1, W2=W2, b2=b2, Wd=Wd, bd=bd)
    for tag, fast in (("fast", True), ("generic", False)):
        h1 = orc.gcn_conv(s, t, n, x, W1, b1, "relu", fast_path=fast, blas=blas)
        h2 = orc.gcn_conv(s, t, n, h1, W2, b2, "relu", fast_path=fast, blas=blas)
        y = orc.matmul(Wd, h2, blas) + bd[None, :]
        c[f"y_{tag}"] = y.astype(np.float32)
        if fast:                                   # hidden activations only for the path the reference takes (size)
            c["h1_fast"], c["h2_fast"] = h1, h2
    return c


def main():
    c = cora_model()
    out = os.path.join(HERE, "config1_cora_v1.npz")
    np.savez_compressed(out, **c)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
