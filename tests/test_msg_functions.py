"""The remaining built-in message functions of GNNlib/src/msgpass.jl:162-208 on the HIP path:
xi_dot_xj, xi_sub_xj, xj_sub_xi through apply_edges (one pass, no gathered temporaries) and e_mul_xj with a MATRIX e
through the fused propagate — against the oracle's gather / scatter composition (the reference's generic path)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def graph(rng, n, E):
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 6, E)      # the last nodes receive nothing
    t[: E // 10] = 3                   # a hub destination (split row)
    p = rng.permutation(E)
    return s[p], t[p]


@pytest.mark.gpu
@pytest.mark.parametrize("D", [1, 7, 64, 100])
@pytest.mark.parametrize("idx", ["int64", "int32"])
def test_two_row_message_functions(gm, oracle, D, idx):
    rng = np.random.default_rng(D)
    n, E = 700, 9000
    s, t = graph(rng, n, E)
    xi = rng.standard_normal((n, D)).astype(np.float32)
    xj = rng.standard_normal((n, D)).astype(np.float32)
    g = gm.GNNGraph(dev(s.astype(idx)), dev(t.astype(idx)), num_nodes=n)
    gi, gj = oracle.gather(xi, t), oracle.gather(xj, s)
    # subtraction of the same two floats: bit-exact
    np.testing.assert_array_equal(gm.apply_edges(gm.xi_sub_xj, g, xi=dev(xi), xj=dev(xj)).cpu().numpy(), gi - gj)
    np.testing.assert_array_equal(gm.apply_edges(gm.xj_sub_xi, g, xi=dev(xi), xj=dev(xj)).cpu().numpy(), gj - gi)
    # sum(xi .* xj, dims = 1): the lanes add in a different order than Julia's sequential sum
    dot = gm.apply_edges(gm.xi_dot_xj, g, xi=dev(xi), xj=dev(xj)).cpu().numpy()
    ref = (gi.astype(np.float64) * gj).sum(1, keepdims=True)
    assert dot.shape == (E, 1)
    assert np.abs(dot - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("aggr", ["+", "mean", "max", "min"])
@pytest.mark.parametrize("D", [5, 32, 100])
def test_propagate_e_mul_xj_matrix(gm, oracle, aggr, D):
    rng = np.random.default_rng(D + len(aggr))
    n, E = 800, 12000
    s, t = graph(rng, n, E)
    x = rng.standard_normal((n, D)).astype(np.float32)
    e = rng.standard_normal((E, D)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    y = gm.propagate(gm.e_mul_xj, g, aggr, xj=dev(x), e=dev(e)).cpu().numpy()
    m = (e * oracle.gather(x, s)).astype(np.float32)           # e .* xj   (msgpass.jl:187-191)
    ref = oracle.scatter(aggr, m, t, n)
    indeg = np.bincount(t - 1, minlength=n)
    short = indeg <= 64
    np.testing.assert_array_equal(y[short], ref[short])        # same products, same order: same bits
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(y), fin)
    assert np.abs(y[fin] - ref[fin]).max() <= 1e-5 * np.abs(ref[fin]).max()
    # and it is the same function as the generic path on materialised messages
    y2 = gm.aggregate_neighbors(g, aggr, dev(m)).cpu().numpy()
    np.testing.assert_array_equal(y2[short], ref[short])
