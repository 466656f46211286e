"""GlobalAttentionPool (GNNlib/src/layers/pool.jl:6-10): α = softmax_nodes(g, fgate(x)); u = reduce_nodes(+, g, α .* ffeat(x)).
GPU path (segment softmax + row multiply + pooled sum, all libgnnmp) against the oracle's composition of the pinned
helpers, and against the closed form per graph in float64."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("gate_channels", [1, 6])
def test_global_attention_pool_vs_oracle(oracle, gate_channels):
    import torch
    import gnnmp
    from gnnmp import synth
    from gnnmp.layers_khop import GlobalAttentionPool
    from oracle import graphwise as GW
    gnnmp.load()
    rng = np.random.default_rng(gate_channels)
    members = synth.batched_graphs(G=300, nmin=5, nmax=40, deg=2, seed=3)
    xs = [rng.standard_normal((n, 10)).astype(np.float32) for _, _, n in members]
    g = gnnmp.batch_arrays(members, xs)
    x = np.concatenate(xs)
    fgate = gnnmp.Dense((10, gate_channels), seed=1)
    ffeat = gnnmp.Dense((10, 6), "relu", seed=2)
    y = GlobalAttentionPool(fgate, ffeat)(g, g.x).cpu().numpy()
    gi = g.graph_indicator.cpu().numpy()
    G = g.num_graphs
    gate = oracle.matmul(fgate.weight.cpu().numpy(), x) + fgate.bias.cpu().numpy()[None, :]
    feats = oracle._act("relu", oracle.matmul(ffeat.weight.cpu().numpy(), x) + ffeat.bias.cpu().numpy()[None, :])
    alpha = GW.softmax_nodes(gi, gate, G)
    ref = oracle.scatter("+", (alpha * feats).astype(np.float32), gi, G)
    assert y.shape == (G, 6)
    assert np.linalg.norm(y - ref) <= 1e-5 * np.linalg.norm(ref)
    # closed form for one graph, float64
    k = 17
    m = gi == k + 1
    a = gate[m].astype(np.float64)
    a = np.exp(a - a.max(0))
    a /= a.sum(0)
    np.testing.assert_allclose(y[k], (a * feats[m]).sum(0), rtol=2e-5, atol=1e-6)
