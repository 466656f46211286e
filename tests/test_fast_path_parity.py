"""the HIP path against the reference's CPU SpMM fast path.

On CPU arrays `propagate(copy_xj | e_mul_xj | w_mul_xj, g, +)` IS `xj * adjacency_matrix(g)` (GNNlib/src/msgpass.jl:215-238)
with the adjacency built by `sparse(s, t, val, n, n)` (GNNGraphs/src/convert.jl:221-237): CSC order (destination columns,
sources ascending inside a column) with duplicate (s, t) entries pre-summed.  That is what "the reference CPU propagate()"
returns for GCNConv, SAGEConv(+) and GraphConv(+); `oracle.spmm_csc` restates it.  The HIP kernels add in original COO
edge order instead, so the comparison is the north_star tolerance (1e-5 relative), not bits.

The graphs carry what makes the two orders differ: duplicate edges (multiplicity folded into one product by the reference),
self loops, hubs that the plan splits into chunks, isolated nodes, edge weights.
"""
import os

import numpy as np
import pytest

RTOL = 1e-5


def _graphs():
    rng = np.random.default_rng(8101)
    out = {}
    # dense multigraph: 60 nodes, 4000 edges => every (s, t) pair about once, many duplicates, self loops included
    n = 60
    out["multi60"] = (rng.integers(1, n + 1, 4000), rng.integers(1, n + 1, 4000), n)
    # a hub whose row is split by the plan (> 64 and > 512 edges), duplicates into the hub, isolated tail nodes
    n = 900
    s = np.concatenate([rng.integers(1, n - 30, 3000), rng.integers(1, 40, 1500), np.arange(1, 701), [9] * 7])
    t = np.concatenate([rng.integers(1, n - 30, 3000), np.full(1500, 5), np.full(700, 6), [9] * 7])
    p = rng.permutation(len(s))
    out["hubs900"] = (s[p], t[p], n)
    # sparse random graph with a few exact duplicates
    n = 2000
    s, t = rng.integers(1, n + 1, 12000), rng.integers(1, n + 1, 12000)
    s[100:160], t[100:160] = s[:60], t[:60]
    out["sparse2000"] = (s, t, n)
    return {k: (np.asarray(a, np.int64), np.asarray(b, np.int64), c) for k, (a, b, c) in out.items()}


GRAPHS = _graphs()


def close(got, ref, rtol=RTOL):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    assert np.linalg.norm(got - ref) <= rtol * np.linalg.norm(ref) + 1e-30        # Julia isapprox(rtol)
    assert np.abs(got - ref).max() <= rtol * np.abs(ref).max() + 1e-30             # element-wise against the array scale


# ---- CPU: the oracle's two paths agree with each other on these graphs (so the GPU comparison below is meaningful) ----
@pytest.mark.parametrize("name", list(GRAPHS))
def test_oracle_fast_path_vs_generic_path(oracle, name):
    s, t, n = GRAPHS[name]
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, 20)).astype(np.float32)
    w = rng.random(len(s)).astype(np.float32)
    close(oracle.spmm_csc(s, t, n, x), oracle.propagate("+", s, t, n, x))
    close(oracle.spmm_csc(s, t, n, x, w), oracle.propagate("+", s, t, n, x, w))
    # duplicates are really folded: an integer-valued input gives exact multiplicities
    xi = np.ones((n, 1), np.float32)
    np.testing.assert_array_equal(oracle.spmm_csc(s, t, n, xi)[:, 0], np.bincount(t - 1, minlength=n).astype(np.float32))


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import gnnmp
    gnnmp.load()
    return gnnmp


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("D", [1, 3, 100, 128])
def test_propagate_sum_vs_reference_spmm(gm, oracle, name, D):
    s, t, n = GRAPHS[name]
    rng = np.random.default_rng(D)
    x = rng.standard_normal((n, D)).astype(np.float32)
    w = rng.random(len(s)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    gw = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=n)
    xd = dev(x)
    # copy_xj: xj * adjacency_matrix(g, weighted = false)                       msgpass.jl:215-218
    close(gm.propagate(gm.copy_xj, g, "+", xj=xd).cpu().numpy(), oracle.spmm_csc(s, t, n, x))
    # ... the weights of a weighted graph are ignored by copy_xj
    close(gm.propagate(gm.copy_xj, gw, "+", xj=xd).cpu().numpy(), oracle.spmm_csc(s, t, n, x))
    # w_mul_xj: xj * adjacency_matrix(g, weighted = true)                        msgpass.jl:234-238
    close(gm.propagate(gm.w_mul_xj, gw, "+", xj=xd).cpu().numpy(), oracle.spmm_csc(s, t, n, x, w))
    # e_mul_xj with a vector e: set_edge_weight(g, e) then the same product      msgpass.jl:224-229
    close(gm.propagate(gm.e_mul_xj, g, "+", xj=xd, e=dev(w)).cpu().numpy(), oracle.spmm_csc(s, t, n, x, w))
    # w_mul_xj on an unweighted graph: adjacency of ones
    close(gm.propagate(gm.w_mul_xj, g, "+", xj=xd).cpu().numpy(), oracle.spmm_csc(s, t, n, x))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("idx", [(1, np.int64), (0, np.int32)])
def test_layers_vs_reference_fast_path(gm, oracle, name, idx):
    """GCNConv (both multiplication orders, weighted and not), SAGEConv(+), GraphConv(+) against the oracle composed with
    the SpMM fast path — what the reference returns on CPU arrays (conv.jl:14-72,102-108,277-283)."""
    base, dt = idx
    s, t, n = GRAPHS[name]
    rng = np.random.default_rng(3)
    Din = 12
    x = rng.standard_normal((n, Din)).astype(np.float32)
    w = (rng.random(len(s)) + 0.1).astype(np.float32)
    sd, td = dev((s - (1 - base)).astype(dt)), dev((t - (1 - base)).astype(dt))
    g = gm.GNNGraph(sd, td, num_nodes=n, index_base=base)
    gw = gm.GNNGraph(sd, td, dev(w), num_nodes=n, index_base=base)
    xd = dev(x)
    for Dout in (20, 5):                                   # Dout >= Din: aggregate first; Dout < Din: weight first
        l = gm.GCNConv((Din, Dout), "relu", seed=4)
        l.bias = dev(rng.standard_normal(Dout).astype(np.float32) * 0.1)
        W, b = l.weight.cpu().numpy(), l.bias.cpu().numpy()
        close(l(g, xd).cpu().numpy(), oracle.gcn_conv(s, t, n, x, W, b, "relu", fast_path=True))
        close(l(g, xd, edge_weight=dev(w)).cpu().numpy(),
              oracle.gcn_conv(s, t, n, x, W, b, "relu", edge_weight=w, fast_path=True))
        lw = gm.GCNConv((Din, Dout), None, use_edge_weight=True, add_self_loops=False, seed=5)
        # isolated destinations have degree 0 => 1/sqrt(0) = Inf and Inf * 0 = NaN in the reference too: compare the finite part
        ref = oracle.gcn_conv(s, t, n, x, lw.weight.cpu().numpy(), lw.bias.cpu().numpy(), None, add_self_loops_=False,
                              use_edge_weight=True, graph_w=w, fast_path=True)
        got = lw(gw, xd).cpu().numpy()
        fin = np.isfinite(ref).all(axis=1)
        np.testing.assert_array_equal(np.isfinite(got).all(axis=1), fin)
        close(got[fin], ref[fin])
    sage = gm.SAGEConv((Din, 20), "relu", aggr="+", seed=6)
    close(sage(g, xd).cpu().numpy(),
          oracle.sage_conv(s, t, n, x, sage.weight.cpu().numpy(), sage.bias.cpu().numpy(), "relu", "+", fast_path=True))
    gc = gm.GraphConv((Din, 20), "relu", aggr="+", seed=7)
    close(gc(g, xd).cpu().numpy(),
          oracle.graph_conv(s, t, n, x, gc.weight1.cpu().numpy(), gc.weight2.cpu().numpy(), gc.bias.cpu().numpy(), "relu",
                            "+", fast_path=True))


@pytest.mark.gpu
def test_arxiv_size_gcn_vs_reference_fast_path(gm, oracle):
    """BASELINE.json config 2 at full size: GCNConv(128 => 128, relu) on the arxiv-shaped graph against the reference's CPU
    path (sparse() rebuild + dense x CSC product per call) — a few seconds of oracle time."""
    from gnnmp import synth
    N, D = synth.ARXIV["N"], synth.ARXIV["D"]
    s, t = synth.arxiv_like()
    x = synth.features(N, D, seed=1)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=N)
    l = gm.GCNConv((D, D), "relu", seed=11)
    got = l(g, dev(x)).cpu().numpy()
    ref = oracle.gcn_conv(s, t, N, x, l.weight.cpu().numpy(), l.bias.cpu().numpy(), "relu", fast_path=True)
    close(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("Din,Dout,sigma,bias", [(40, 8, "relu", True), (1433, 64, "relu", True), (100, 64, None, True),
                                                  (33, 7, "relu", False), (16, 4, None, False)])
def test_gcn_w_first_epilogue_in_the_row_kernel(gm, oracle, Din, Dout, sigma, bias):
    """Dout < Din: `x = W * x` before the convolution, `σ.(x .+ b)` after it (conv.jl:36-40,71).  The bias and the relu ride in
    the row kernel (gnnmp_propagate_slots_act_f32): bit-identical to propagate_slots + bias_act, within 1e-5 of the oracle —
    hubs above the split threshold included (their rows are finished by the combine kernel)."""
    import torch
    from gnnmp import _lib as L
    from gnnmp.layers import bias_act, dense, gcn_norm_cache
    rng = np.random.default_rng(Din + Dout)
    n, E = 2000, 30000
    s = rng.integers(1, n + 1, E)
    t = np.concatenate([rng.integers(1, n + 1, E - 900), np.full(900, 7)])
    x = rng.standard_normal((n, Din)).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    l = gm.GCNConv((Din, Dout), sigma, bias=bias, seed=3)
    if bias:
        l.bias = dev((rng.standard_normal(Dout) * 0.3).astype(np.float32))
    y = l(g, dev(x))
    ref = oracle.gcn_conv(s, t, n, x, l.weight.cpu().numpy(), l.bias.cpu().numpy() if bias else None, sigma)
    got = y.cpu().numpy()
    assert np.linalg.norm(got - ref) <= 1e-5 * np.linalg.norm(ref)
    # the unfused composition on the same plan: same bits
    lib = L.load()
    plan = g.plan(True)
    c, c_slot, _ = gcn_norm_cache(g, True)
    h = dense(dev(x), l.weight)
    agg = torch.empty_like(h)
    L.check(lib.gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(h), None, L.ptr(c_slot), L.ptr(c), L.ptr(agg), Dout,
                                          L.stream_ptr()))
    assert torch.equal(y, bias_act(agg, l.bias, sigma))
