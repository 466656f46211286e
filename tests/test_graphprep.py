"""Device-side graph prep: sort_edge_index, is_bidirected, has_self_loops, sample_neighbors.

Index work: bit-exact against the reference definition (`sortperm(collect(zip(u, v)))`, utils.jl:41-45).  The sampler
cannot share Julia's RNG stream, so it is held to the properties the reference's tests assert
(GNNGraphs/test/sampling.jl:3-33: sizes, EID uniqueness, membership, neighbour sets) plus a uniformity check."""
import numpy as np
import pytest


def ref_sort_edge_index(u, v):
    """utils.jl:41-45 verbatim in python: sortperm of the tuples"""
    uv = sorted(zip(u.tolist(), v.tolist()))
    return np.array([a for a, _ in uv], dtype=u.dtype), np.array([b for _, b in uv], dtype=v.dtype)


def test_reference_definition_equals_numpy_lexsort():
    rng = np.random.default_rng(0)
    u = rng.integers(1, 50, 2000)
    v = rng.integers(1, 50, 2000)
    a, b = ref_sort_edge_index(u, v)
    p = np.lexsort((v, u))
    assert (a == u[p]).all() and (b == v[p]).all()


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


@pytest.mark.gpu
@pytest.mark.parametrize("idx", ["int64", "int32"])
@pytest.mark.parametrize("n,E", [(10, 40), (1000, 50000), (3_000_000, 200000), (5, 1)])
def test_sort_edge_index_bit_exact(gm, idx, n, E):
    from gnnmp import sampling as S
    rng = np.random.default_rng(E)
    u = rng.integers(1, n + 1, E).astype(idx)
    v = rng.integers(1, n + 1, E).astype(idx)
    uo, vo = S.sort_edge_index(dev(u), dev(v))
    p = np.lexsort((v, u))
    np.testing.assert_array_equal(uo.cpu().numpy(), u[p])
    np.testing.assert_array_equal(vo.cpu().numpy(), v[p])
    uo2, vo2 = S.sort_edge_index((dev(u), dev(v)))             # tuple form (utils.jl:30)
    assert bool((uo2 == uo).all()) and bool((vo2 == vo).all())


@pytest.mark.gpu
def test_sort_edge_index_rejects_bad_indices(gm):
    from gnnmp import _lib as L, sampling as S
    with pytest.raises(L.GnnmpError) as ei:
        S.sort_edge_index(dev(np.array([1, 0, 3])), dev(np.array([1, 2, 3])))     # 0 is not a 1-based index
    assert ei.value.status == L.EBOUNDS


@pytest.mark.gpu
def test_is_bidirected_and_has_self_loops(gm):
    from gnnmp import sampling as S
    rng = np.random.default_rng(1)
    a = rng.integers(1, 200, 3000)
    b = rng.integers(1, 200, 3000)
    keep = a != b
    a, b = a[keep], b[keep]
    g_dir = gm.GNNGraph(dev(a), dev(b), num_nodes=200)
    g_bi = gm.GNNGraph(dev(np.concatenate([a, b])), dev(np.concatenate([b, a])), num_nodes=200)
    s1, t1 = ref_sort_edge_index(a, b)
    s2, t2 = ref_sort_edge_index(b, a)
    assert S.is_bidirected(g_dir) == bool(((s1 == s2) & (t1 == t2)).all())     # query.jl:553-558
    assert S.is_bidirected(g_bi) is True
    assert S.has_self_loops(g_bi) is False
    g_loop = gm.GNNGraph(dev(np.append(a, 7)), dev(np.append(b, 7)), num_nodes=200)
    assert S.has_self_loops(g_loop) is True
    assert S.has_self_loops(gm.add_self_loops(g_dir)) is True


def _adj_eids(s, t, n, dir):
    """adjacency_list(g; dir, with_eid = true) — query.jl: per node, the positions (1-based) of its in / out edges"""
    key = t if dir == "in" else s
    return [np.nonzero(key == i)[0] + 1 for i in range(1, n + 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("idx", ["int64", "int32"])
def test_sample_neighbors_reference_properties(gm, idx):
    from gnnmp import sampling as S
    rng = np.random.default_rng(4)
    n, E = 10, 40                                                   # test/sampling.jl:6
    s = rng.integers(1, n + 1, E).astype(idx)
    t = rng.integers(1, n + 1, E).astype(idx)
    g = gm.GNNGraph(dev(s), dev(t), num_nodes=n)
    nodes = np.array([2, 3])
    # replace = false, K = -1, dir = :in   (sampling.jl:3-17)
    sg = S.sample_neighbors(g, dev(nodes), dir="in")
    adj = _adj_eids(s, t, n, "in")
    eid = sg.eid.cpu().numpy()
    assert sg.num_nodes == n
    assert sg.num_edges == sum(len(adj[i - 1]) for i in nodes) == len(eid)
    assert len(np.unique(eid)) == len(eid)
    ss, tt = sg.s.cpu().numpy(), sg.t.cpu().numpy()
    assert np.isin(tt, nodes).all()
    np.testing.assert_array_equal(ss, s[eid - 1])
    np.testing.assert_array_equal(tt, t[eid - 1])
    for i in nodes:
        assert sorted(ss[tt == i]) == sorted(s[t == i])
    # replace = true, K = 2, dir = :out    (sampling.jl:19-33)
    sg = S.sample_neighbors(g, dev(nodes), 2, dir="out", replace=True, seed=5)
    adjo = _adj_eids(s, t, n, "out")
    has_out = [i for i in nodes if len(adjo[i - 1]) > 0]
    assert sg.num_edges == 2 * len(has_out)
    ss, tt, eid = sg.s.cpu().numpy(), sg.t.cpu().numpy(), sg.eid.cpu().numpy()
    assert np.isin(ss, nodes).all()
    for i in nodes:
        assert set(tt[ss == i]) <= set(t[s == i])
        assert set(eid[ss == i]) <= set(adjo[i - 1])


def _ref_dropnodes(s, t, nodes, eids, dir):
    """sampling.jl:100-116 restated: nodes_all = [nodes; setdiff(s | t, nodes)], edges relabelled by position"""
    ss, tt = s[eids - 1], t[eids - 1]
    other = []
    seen = set(int(v) for v in nodes)
    for v in (ss if dir == "in" else tt):                            # setdiff keeps first-appearance order
        if int(v) not in seen:
            seen.add(int(v))
            other.append(int(v))
    nodes_all = np.array([int(v) for v in nodes] + other)
    pos = {v: i + 1 for i, v in enumerate(nodes_all)}
    return nodes_all, np.array([pos[int(v)] for v in ss]), np.array([pos[int(v)] for v in tt])


@pytest.mark.gpu
@pytest.mark.parametrize("dir,K,idx", [("in", -1, "int64"), ("out", -1, "int32"), ("in", 3, "int64"), ("out", 2, "int64")])
def test_sample_neighbors_dropnodes_relabels_like_the_reference(gm, dir, K, idx):
    from gnnmp import sampling as S
    rng = np.random.default_rng(12)
    n, E, D = 60, 500, 5
    s = rng.integers(1, n + 1, E).astype(idx)
    t = rng.integers(1, n + 1, E).astype(idx)
    x = rng.standard_normal((n, D)).astype(np.float32)
    w = rng.random(E).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=n, x=dev(x))
    nodes = np.array([7, 2, 31, 40])
    sg = S.sample_neighbors(g, dev(nodes), K, dir=dir, dropnodes=True, seed=3)
    eid = sg.eid.cpu().numpy()
    nid = sg.nid.cpu().numpy()
    nodes_all, rs, rt = _ref_dropnodes(s, t, nodes, eid, dir)       # the relabelling is deterministic given the draw
    np.testing.assert_array_equal(nid, nodes_all)
    np.testing.assert_array_equal(sg.s.cpu().numpy(), rs)
    np.testing.assert_array_equal(sg.t.cpu().numpy(), rt)
    assert sg.s.dtype == g.s.dtype and sg.num_nodes == len(nodes_all)
    # test/sampling.jl:35-46
    if K < 0:
        key = t if dir == "in" else s
        assert sg.num_edges == sum(int((key == i).sum()) for i in nodes)
    assert eid.shape == (sg.num_edges,) and nid.shape == (sg.num_nodes,)
    np.testing.assert_array_equal(sg.w.cpu().numpy(), w[eid - 1])
    np.testing.assert_array_equal(sg.x.cpu().numpy(), x[nid - 1])
    assert len(np.unique(nid)) == len(nid)
    # the relabelled graph is usable on the path: propagate over it equals the oracle on the relabelled COO
    out = gm.propagate(gm.copy_xj, sg, "+", xj=sg.x).cpu().numpy()
    want = np.zeros((len(nid), D), np.float32)
    for a, b in zip(rs, rt):
        want[b - 1] += x[nid[a - 1] - 1]
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_sample_neighbors_large_counts_uniqueness_uniformity(gm):
    from gnnmp import sampling as S
    rng = np.random.default_rng(8)
    n, E, K = 5000, 120000, 5
    s = rng.integers(1, n + 1, E)
    t = rng.integers(1, n - 50, E)                                  # the last nodes have no incoming edge
    t[:4000] = 9                                                    # a hub
    w = rng.random(E).astype(np.float32)
    g = gm.GNNGraph(dev(s), dev(t), dev(w), num_nodes=n)
    nodes = rng.permutation(n)[:2000] + 1
    indeg = np.bincount(t - 1, minlength=n)
    sg = S.sample_neighbors(g, dev(nodes), K, seed=3)
    off = sg.sample_offsets.cpu().numpy()
    np.testing.assert_array_equal(np.diff(off), np.minimum(indeg[nodes - 1], K))
    eid = sg.eid.cpu().numpy()
    assert len(np.unique(eid)) == len(eid)                          # without replacement: no edge twice
    np.testing.assert_array_equal(t[eid - 1], np.repeat(nodes, np.diff(off)))
    np.testing.assert_array_equal(sg.w.cpu().numpy(), w[eid - 1])   # weights follow their edges (sampling.jl:84-86)
    # same seed, same sample; another seed, another sample
    assert bool((S.sample_neighbors(g, dev(nodes), K, seed=3).eid == sg.eid).all())
    assert not bool((S.sample_neighbors(g, dev(nodes), K, seed=4).eid == sg.eid).all())
    # uniformity on the hub: each of its d edges should be kept with probability K/d; pool 400 seeds -> chi-square-ish
    hub = np.array([9])
    d = indeg[8]
    hits = np.zeros(E + 1)
    R = 400
    for r in range(R):
        e = S.sample_neighbors(g, dev(hub), 200, seed=1000 + r).eid.cpu().numpy()
        assert len(e) == 200 and len(np.unique(e)) == 200
        hits[e] += 1
    h = hits[np.nonzero(t == 9)[0] + 1]
    assert h.sum() == 200 * R and (hits.sum() == h.sum())
    p = 200 / d
    z = (h - R * p) / np.sqrt(R * p * (1 - p))
    assert abs(z.mean()) < 0.15 and 0.85 < z.std() < 1.15 and np.abs(z).max() < 5.5
    # few picks from the hub (d > 2 K^2: the kernel's Floyd branch instead of the O(d) selection scan): same properties
    hits[:] = 0
    R, Kf = 3000, 8
    hub_pos = np.nonzero(t == 9)[0] + 1
    for r in range(R):
        e = S.sample_neighbors(g, dev(hub), Kf, seed=50000 + r).eid.cpu().numpy()
        assert len(e) == Kf and len(np.unique(e)) == Kf
        assert (np.diff(np.searchsorted(hub_pos, e)) > 0).all()     # kept in original edge order
        hits[e] += 1
    h = hits[hub_pos]
    assert h.sum() == Kf * R and hits.sum() == h.sum()
    p = Kf / d
    z = (h - R * p) / np.sqrt(R * p * (1 - p))
    assert abs(z.mean()) < 0.15 and 0.85 < z.std() < 1.15 and np.abs(z).max() < 6.0
    # with replacement: k independent uniform picks
    e = S.sample_neighbors(g, dev(hub), 20000, replace=True, seed=2).eid.cpu().numpy()
    assert len(e) == 20000 and set(e) <= set(np.nonzero(t == 9)[0] + 1)
    cnt = np.bincount(e, minlength=E + 1)[np.nonzero(t == 9)[0] + 1]
    assert abs(cnt.mean() - 20000 / d) < 1e-9 and cnt.std() < 3.0 * np.sqrt(20000 / d)


@pytest.mark.gpu
@pytest.mark.parametrize("E", [1, 2, 63, 64, 65, 2047, 2048, 2049, 4097, 70001, 300007])
@pytest.mark.parametrize("n", [1, 3, 255, 256, 257, 70000])
def test_sort_and_scan_primitives_at_tile_boundaries(gm, E, n):
    """the hand-written radix sort / scan (csrc/sort_scan.hip) through their callers, at sizes around the 64-lane step, the
    2 048-element wave tile, the 4 096-element scan chunk and the 8-bit digit: the plan's stable destination sort against
    numpy's stable argsort (bit-exact index outputs), sort_edge_index against numpy's lexsort"""
    import torch
    rng = np.random.default_rng(E * 7 + n)
    s = rng.integers(1, n + 1, E).astype(np.int64)
    t = rng.integers(1, n + 1, E).astype(np.int64)
    g = gm.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=n)
    for loops in (False, True):
        rowptr, col, eid = (v.cpu().numpy() for v in g.plan(loops).export())
        s2 = np.concatenate([s, np.arange(1, n + 1)]) if loops else s
        t2 = np.concatenate([t, np.arange(1, n + 1)]) if loops else t
        order = np.argsort(t2, kind="stable")
        np.testing.assert_array_equal(eid, order.astype(np.int32))
        np.testing.assert_array_equal(col, (s2[order] - 1).astype(np.int32))
        np.testing.assert_array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(t2 - 1, minlength=n))]).astype(np.int32))
    a, b = gm.sort_edge_index(g.s, g.t)
    o = np.lexsort((t, s))
    np.testing.assert_array_equal(a.cpu().numpy(), s[o])
    np.testing.assert_array_equal(b.cpu().numpy(), t[o])
