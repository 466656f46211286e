"""reduce_nodes / GlobalPool when the batch holds few, LARGE graphs (a whole-graph readout is num_graphs = 1): the segments
must be reduced in parallel chunks, not by one lane group walking millions of rows (283 ms at N = 2.4 M before the fix).
Values against float64 numpy; max / min exactly; and a loose wall-clock bound that the serial walk could not meet."""
import time

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("G", [1, 2, 3])
@pytest.mark.parametrize("form", ["graph", "indicator"])
def test_reduce_nodes_with_large_segments(oracle, G, form):
    import torch
    import gnnmp
    gnnmp.load()
    rng = np.random.default_rng(G)
    N, D = 600_000, 20
    cuts = np.sort(rng.choice(np.arange(1, N), G - 1, replace=False)) if G > 1 else np.array([], np.int64)
    gi = np.searchsorted(cuts, np.arange(N), side="right") + 1
    x = rng.standard_normal((N, D)).astype(np.float32)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    X, GI = dev(x), dev(gi)
    g = gnnmp.GNNGraph(dev(np.array([1, 2])), dev(np.array([2, 1])), num_nodes=N, graph_indicator=GI, num_graphs=G)
    arg = g if form == "graph" else GI
    kw = {} if form == "graph" else {"num_graphs": G}
    for aggr in ("+", "mean", "max", "min"):
        y = gnnmp.reduce_nodes(aggr, arg, X, **kw).cpu().numpy()
        assert y.shape == (G, D)
        for k in range(G):
            seg = x[gi == k + 1].astype(np.float64)
            ref = {"+": seg.sum(0), "mean": seg.mean(0), "max": seg.max(0), "min": seg.min(0)}[aggr]
            if aggr in ("max", "min"):
                np.testing.assert_array_equal(y[k], ref.astype(np.float32))
            else:
                assert np.abs(y[k] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()) * (30 if aggr == "+" else 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gnnmp.reduce_nodes("mean", arg, X, **kw)
    torch.cuda.synchronize()
    assert (time.perf_counter() - t0) / 5 < 0.03          # the one-lane-group walk took ~70 ms at this size


@pytest.mark.gpu
def test_global_pool_layer_on_one_large_graph():
    import torch
    import gnnmp
    gnnmp.load()
    N, D = 300_000, 16
    x = torch.randn((N, D), device="cuda")
    g = gnnmp.GNNGraph(torch.tensor([1, 2]).cuda(), torch.tensor([2, 1]).cuda(), num_nodes=N)
    y = gnnmp.GlobalPool("mean")(g, x)
    assert y.shape == (1, D)
    assert torch.allclose(y[0].double(), x.double().mean(0), atol=1e-5)
