"""The one-pass narrow-row softmax (csrc/softmax_rows.hip) behind softmax_edge_neighbors (GNNlib/src/utils.jl:84-97),
softmax_nodes and softmax_edges (:49-72):
  * BIT-IDENTICAL to the three-step kernels it replaces on rows the plan does not split (knob 16 < 0 = three steps on every
    row) — same max, same exponentials, same edge-order sum, same division;
  * within 1e-6 of the oracle's restatement of the reference;
  * hubs above the split threshold (three-step path on those rows only), empty destinations, rows of every width the kernel
    takes (1 .. 16 channels) and one it does not (fallback), single-row and single-edge graphs, eps in the denominator."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    yield gnnmp
    gnnmp.tune(16, 0)


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hub_graph(rng, n, E, hubs):
    s = [rng.integers(0, n, E)]
    t = [rng.integers(0, max(1, n - 30), E)]            # the last 30 destinations stay empty
    for k, m in enumerate(hubs):
        s.append(rng.integers(0, n, m))
        t.append(np.full(m, min(n - 1, 5 + 11 * k)))
    s, t = np.concatenate(s), np.concatenate(t)
    p = rng.permutation(len(s))
    return s[p].astype(np.int64) + 1, t[p].astype(np.int64) + 1


def both_paths(gm, fn):
    import torch
    gm.tune(16, 0)
    new = fn()
    gm.tune(16, -1)
    old = fn()
    gm.tune(16, 0)
    torch.cuda.synchronize()
    return new, old


@pytest.mark.parametrize("H", [1, 2, 3, 4, 6, 8, 16, 40])
@pytest.mark.parametrize("n,E,hubs", [(900, 20000, (700, 90, 65)), (70, 300, ()), (5000, 90000, (1400, 600))])
def test_edge_softmax_matches_three_steps_and_oracle(gm, oracle, H, n, E, hubs):
    import torch
    rng = np.random.default_rng(H * 1009 + n)
    s, t = hub_graph(rng, n, E, hubs)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    e = (rng.standard_normal((len(s), H)) * 3).astype(np.float32)
    ed = dev(e)
    new, old = both_paths(gm, lambda: gm.softmax_edge_neighbors(g, ed))
    assert torch.equal(new, old)
    assert torch.equal(new, gm.softmax_edge_neighbors(g, ed))       # no atomics anywhere: run-to-run identical
    ref = oracle.softmax_edge_neighbors(t, n, e)                    # 1-based, like the graph
    np.testing.assert_allclose(new.cpu().numpy(), ref, rtol=5e-6, atol=1e-12)   # split rows fold chunk partials: order differs from the oracle
    # every destination with an edge sums to one
    z = np.zeros((n, H), np.float64)
    np.add.at(z, t - 1, new.cpu().numpy().astype(np.float64))
    has = np.bincount(t - 1, minlength=n) > 0
    np.testing.assert_allclose(z[has], 1.0, rtol=1e-5)


def test_rows_exactly_at_the_batch_and_split_boundaries(gm, oracle):
    """row lengths around the slot capacity of a batch (512 at H = 8) and around the plan's split threshold (info[7])"""
    import torch
    rng = np.random.default_rng(5)
    lens = [0, 1, 63, 64, 65, 255, 256, 257, 448, 511, 512, 513, 1, 0, 0, 7, 500, 12, 12, 500, 3]
    t = np.concatenate([np.full(m, i) for i, m in enumerate(lens)]).astype(np.int64)
    n = len(lens) + 70                                       # more than one wave of destinations
    t = np.concatenate([t, rng.integers(len(lens), n, 4000)])
    s = rng.integers(0, n, len(t)).astype(np.int64)
    p = rng.permutation(len(t))
    s, t = s[p], t[p]
    for H in (8, 1, 4):
        e = (rng.standard_normal((len(t), H)) * 2).astype(np.float32)
        for thresh in (0, 512, 128):                         # 0 = the plan's own choice (64 at this size)
            gm.tune(4, thresh)
            try:
                g = gm.GNNGraph(dev(s + 1), dev(t + 1), num_nodes=n)
                new, old = both_paths(gm, lambda: gm.softmax_edge_neighbors(g, dev(e)))
            finally:
                gm.tune(4, 0)
            assert torch.equal(new, old)
            np.testing.assert_allclose(new.cpu().numpy(), oracle.softmax_edge_neighbors(t + 1, n, e), rtol=2e-6, atol=1e-12)


def test_extreme_logits_and_ties(gm, oracle):
    import torch
    rng = np.random.default_rng(9)
    n, E = 300, 6000
    s, t = hub_graph(rng, n, E, (200,))
    e = rng.standard_normal((len(s), 8)).astype(np.float32)
    e[::7] = 80.0
    e[1::11] = -90.0
    e[2::13] = e[3::13][: len(e[2::13])]                     # exact ties
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    new, old = both_paths(gm, lambda: gm.softmax_edge_neighbors(g, dev(e)))
    assert torch.equal(new, old)
    assert torch.isfinite(new).all()
    np.testing.assert_allclose(new.cpu().numpy(), oracle.softmax_edge_neighbors(t, n, e), rtol=2e-6, atol=1e-30)


def test_segment_softmax_eps_denominator(gm):
    """softmax_edges adds eps(T) to the denominator (utils.jl:71); softmax_nodes does not: both through the one-pass kernel"""
    import torch
    from gnnmp import utils as U
    rng = np.random.default_rng(3)
    parts = []
    for k in range(9):
        nk = int(rng.integers(3, 40))
        mk = int(rng.integers(1, 200))
        parts.append((rng.integers(0, nk, mk), rng.integers(0, nk, mk), nk))
    g = gm.batch([gm.GNNGraph(dev(a.astype(np.int64) + 1), dev(b.astype(np.int64) + 1), num_nodes=nk) for a, b, nk in parts])
    x = rng.standard_normal((g.num_nodes, 5)).astype(np.float32)
    e = rng.standard_normal((g.num_edges, 2)).astype(np.float32)
    for fn, arr in ((U.softmax_nodes, x), (U.softmax_edges, e)):
        new, old = both_paths(gm, lambda: fn(g, dev(arr)))
        assert torch.equal(new, old)


def test_one_destination_one_edge_and_no_edges(gm):
    import torch
    g = gm.GNNGraph(dev(np.array([1], np.int64)), dev(np.array([1], np.int64)), num_nodes=1)
    z = gm.softmax_edge_neighbors(g, dev(np.array([[3.0, -2.0]], np.float32)))
    assert torch.equal(z.cpu(), torch.ones(1, 2))
    g0 = gm.GNNGraph(dev(np.zeros(0, np.int64)), dev(np.zeros(0, np.int64)), num_nodes=4)
    assert gm.softmax_edge_neighbors(g0, torch.empty((0, 8), device="cuda")).shape == (0, 8)


def test_nan_logit_poisons_its_destination_only(gm, oracle):
    """the kernels fold the maximum with v_max_f32 (a NaN is skipped, not carried): the weights must still be what the
    reference's NaN-carrying max gives — NaN on every edge of the destination (and channel) that holds the NaN, untouched
    elsewhere — on batch rows, rows longer than a batch and rows the plan splits"""
    import torch
    rng = np.random.default_rng(21)
    n, E = 400, 9000
    s, t = hub_graph(rng, n, E, (300, 700))
    e = rng.standard_normal((len(s), 8)).astype(np.float32)
    bad_edges = [3, 17, int(np.flatnonzero(t == 6)[5]), int(np.flatnonzero(t == 17)[100])]   # 6, 17: the hubs
    for k, ed in enumerate(bad_edges):
        e[ed, k % 8] = np.nan
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    new, old = both_paths(gm, lambda: gm.softmax_edge_neighbors(g, dev(e)))
    ref = oracle.softmax_edge_neighbors(t, n, e)
    got = new.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(np.isnan(old.cpu().numpy()), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=2e-6, atol=1e-30)
    assert np.isnan(ref).sum() >= len(bad_edges)
