"""The per-step preparation of a graph-classification loop (examples/graph_classification_tudataset.jl:70-71, 97-104: a NEW batch every
step) without a sort and without a host synchronisation:
  gnnmp_plan_select / gnnmp_plan_concat   the batch's plan from a resident dataset plan / from the members' plans — bit-identical
                                          (rowptr, col, eid) to gnnmp_plan_create on MLUtils.batch's COO (transform.jl:682-709)
  gnnmp_chain_jobs_pack                   the fused chain's wave jobs packed on the device — a valid packing, as tight as the host's best fit
                                          decreasing, and the chain's logits bit-identical with either
  gnnmp.GraphDataset / DataLoader         the host mirror that strings them together
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def random_members(G, rng, nmin=1, nmax=40, hub=0):
    out = []
    for _ in range(G):
        n = int(rng.integers(nmin, nmax + 1))
        m = int(rng.integers(0, 4 * n + 1))
        s = rng.integers(0, n, m)
        t = rng.integers(0, n, m)
        if hub and rng.random() < 0.2:          # one node with `hub` extra in-edges (multi-edges allowed: COO_T is a multigraph)
            s = np.concatenate([s, rng.integers(0, n, hub)])
            t = np.concatenate([t, np.full(hub, int(rng.integers(0, n)))])
        out.append((s.astype(np.int64) + 1, t.astype(np.int64) + 1, n))
    return out


def exported(plan):
    import torch
    return tuple(a.cpu().numpy() for a in plan.export())


def same_plan(p, q, tag):
    assert (p.n_dst, p.n_edges, p.n_total) == (q.n_dst, q.n_edges, q.n_total), tag
    for name, a, b in zip(("rowptr", "col", "eid"), exported(p), exported(q)):
        assert np.array_equal(a, b), f"{tag}: {name} differs"


@pytest.mark.parametrize("seed,k", [(0, 1), (1, 7), (2, 300), (3, 2500)])
def test_select_is_plan_create_on_the_batched_coo(gm, seed, k):
    import torch
    from gnnmp import _lib as L
    rng = np.random.default_rng(seed)
    members = random_members(400, rng)
    xs = [rng.standard_normal((n, 5), dtype=np.float32) for _, _, n in members]
    ds = gm.GraphDataset.from_members(members, xs)
    ids = rng.integers(0, len(members), k)            # with repeats: batch(gs[ids]) takes any index vector
    g = ds.batch(ids)
    g._plans[False].status()
    ref = gm.batch_arrays([members[i] for i in ids], [xs[i] for i in ids])
    same_plan(g.plan(False), ref.plan(False), "select vs plan_create")
    assert torch.equal(g.x, ref.x)
    assert torch.equal(g.graph_indicator, ref.graph_indicator)
    sizes = np.array([members[i][2] for i in ids])
    assert np.array_equal(g._cache["node_ptr"].cpu().numpy(), np.concatenate([[0], np.cumsum(sizes)]))
    # the node map is getgraph's nmap (transform.jl:871-873): batch row -> dataset row
    nptr = np.concatenate([[0], np.cumsum([m[2] for m in members])])
    want = np.concatenate([np.arange(nptr[i], nptr[i + 1]) for i in ids]) if k else np.zeros(0)
    assert np.array_equal(g._cache["node_map"].cpu().numpy(), want)
    # s, t materialised from the plan = MLUtils.batch's COO
    assert torch.equal(g.s, ref.s) and torch.equal(g.t, ref.t)
    assert (g.num_nodes, g.num_edges, g.num_graphs) == (ref.num_nodes, ref.num_edges, ref.num_graphs)
    # with plan-added self loops (the dataset plan built with add_self_loops): the loops stay at E_batch + node
    lib = L.load()
    h = ctypes.c_void_p()
    ids_dev = torch.from_numpy(ids + 1).cuda()
    L.check(lib.gnnmp_plan_select(ctypes.byref(h), ds.gall.plan(True).handle, L.ptr(ds.node_ptr), len(members), L.ptr(ids_dev), 8, 1, k,
                                  ref.num_nodes, ref.num_edges + ref.num_nodes, None, None, None, L.stream_ptr()))
    from gnnmp.graph import Plan
    same_plan(Plan._adopt(h, g.device), ref.plan(True), "select vs plan_create, self loops")


@pytest.mark.parametrize("loops", [False, True])
def test_concat_is_plan_create_on_the_batched_coo(gm, loops):
    import torch
    from gnnmp.graph import Plan
    rng = np.random.default_rng(11)
    members = random_members(120, rng, nmin=1, nmax=30)
    members[3] = (np.zeros(0, np.int64), np.zeros(0, np.int64), 4)          # a member without edges
    members[7] = (np.array([1], np.int64), np.array([1], np.int64), 1)      # one node, one self edge
    plans = []
    for s, t, n in members:
        plans.append(Plan(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), n, n, 1, loops, validate=True))
    order = rng.permutation(len(members))
    plan, seg, gi = gm.concat_plans([plans[i] for i in order], want_indicator=True)
    ref = gm.batch_arrays([members[i] for i in order])
    same_plan(plan, ref.plan(loops), f"concat vs plan_create (self loops {loops})")
    assert torch.equal(gi, ref.graph_indicator)
    assert np.array_equal(seg.cpu().numpy(), np.concatenate([[0], np.cumsum([members[i][2] for i in order])]))
    # and it computes: propagate on the concatenated plan == on the sorted one, bit for bit
    x = torch.randn((ref.num_nodes, 12), device="cuda")
    g2 = gm.GNNGraph._from_plan(plan, len(order), gi, x, 1, torch.int64, self_loops=loops)
    if not loops:
        y = gm.propagate(gm.copy_xj, g2, "+", xj=x)
        assert torch.equal(y, gm.propagate(gm.copy_xj, ref, "+", xj=x))


def test_hub_rows_get_their_chunk_tables(gm):
    """a member with a destination of more than GNNMP_MIN_LONG_ROW in-edges: the new plan needs the split-row tables (the one case in
    which the call synchronises) — same split, same results as the sorted plan"""
    import torch
    rng = np.random.default_rng(5)
    members = random_members(60, rng, nmin=8, nmax=40, hub=150)
    xs = [rng.standard_normal((n, 8), dtype=np.float32) for _, _, n in members]
    ds = gm.GraphDataset.from_members(members, xs)
    ids = rng.permutation(len(members))[:50]
    g = ds.batch(ids)
    ref = gm.batch_arrays([members[i] for i in ids], [xs[i] for i in ids])
    p, q = g.plan(False), ref.plan(False)
    assert q.n_long > 0 and p.n_long == q.n_long and p.max_degree == q.max_degree
    same_plan(p, q, "hub batch")
    for aggr in ("+", "mean", "max"):
        assert torch.equal(gm.propagate(gm.copy_xj, g, aggr, xj=g.x), gm.propagate(gm.copy_xj, ref, aggr, xj=ref.x)), aggr


def test_wrong_totals_give_an_empty_plan_and_a_status(gm):
    import torch
    from gnnmp import _lib as L
    from gnnmp.graph import Plan
    rng = np.random.default_rng(8)
    members = random_members(50, rng)
    ds = gm.GraphDataset.from_members(members)
    ids = np.arange(10)
    n_rows = int(sum(members[i][2] for i in ids))
    n_edges = int(sum(len(members[i][0]) for i in ids))
    lib = L.load()
    for bad_rows, bad_edges, ids_dev, what in ((n_rows + 1, n_edges, torch.from_numpy(ids + 1).cuda(), "totals"),
                                               (n_rows, n_edges, torch.from_numpy(ids + 1000).cuda(), "id range")):
        h = ctypes.c_void_p()
        L.check(lib.gnnmp_plan_select(ctypes.byref(h), ds.plan.handle, L.ptr(ds.node_ptr), len(members), L.ptr(ids_dev), 8, 1, len(ids),
                                      bad_rows, bad_edges, None, None, None, L.stream_ptr()))
        p = Plan._adopt(h, ds.device)
        with pytest.raises(L.GnnmpError):
            p.status()
        rowptr, _, _ = p.export()
        assert int(rowptr.abs().sum()) == 0, what      # nothing points anywhere


def test_pooled_blocks_are_reused_across_streams(gm):
    """create -> use -> release in a loop, alternating streams: the block of a released plan is handed out again behind an event (no host
    synchronisation), and every batch still equals the sorted plan"""
    import torch
    rng = np.random.default_rng(21)
    members = random_members(300, rng)
    xs = [rng.standard_normal((n, 4), dtype=np.float32) for _, _, n in members]
    ds = gm.GraphDataset.from_members(members, xs)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for it in range(12):
        ids = rng.integers(0, len(members), 200)
        with torch.cuda.stream(streams[it & 1]):
            g = ds.batch(ids)
            y = gm.propagate(gm.copy_xj, g, "+", xj=g.x)
            del g                                    # released on this stream, behind the propagate
        ref = gm.batch_arrays([members[i] for i in ids], [xs[i] for i in ids])
        yr = gm.propagate(gm.copy_xj, ref, "+", xj=ref.x)
        torch.cuda.synchronize()
        assert torch.equal(y, yr), it


def test_pool_is_keyed_on_the_device_and_release_follows_the_last_use(gm):
    """(i) a block parked on device 0 is not handed to a taker whose current device is another one (mocked through the library's test
    hook: the box has one GPU), and comes back to device 0's next taker; (ii) a pooled plan is released on the stream it was LAST USED on,
    not on whatever stream is current when the garbage collector runs"""
    import torch
    from gnnmp import _lib as L
    lib = L.load()
    lib.gnnmp_debug_mock_device.argtypes = [ctypes.c_int]
    lib.gnnmp_debug_plan_block.restype = ctypes.c_void_p
    lib.gnnmp_debug_plan_block.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(5)
    members = random_members(200, rng)
    xs = [rng.standard_normal((n, 4), dtype=np.float32) for _, _, n in members]
    ds = gm.GraphDataset.from_members(members, xs)
    ids = rng.integers(0, len(members), 150)
    torch.cuda.synchronize()
    try:
        g = ds.batch(ids, with_x=False)
        blk0 = lib.gnnmp_debug_plan_block(g.plan().handle)
        assert blk0
        del g                                               # parked in device 0's table
        assert lib.gnnmp_debug_mock_device(1) == 1
        g1 = ds.batch(ids, with_x=False)
        blk1 = lib.gnnmp_debug_plan_block(g1.plan().handle)
        assert blk1 and blk1 != blk0, "a block parked on device 0 was handed out under device 1"
        lib.gnnmp_debug_mock_device(-1)
        g2 = ds.batch(ids, with_x=False)
        assert lib.gnnmp_debug_plan_block(g2.plan().handle) == blk0      # same size, same device: the parked block
        del g1, g2
    finally:
        lib.gnnmp_debug_mock_device(-1)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(a):
        g = ds.batch(ids)
        y = gm.propagate(gm.copy_xj, g, "+", xj=g.x)
    assert g.plan()._last_stream == a
    with torch.cuda.stream(b):
        plan = g.plan()
        del g
        assert plan._last_stream == a                      # reading attributes does not move it; only compute calls do
        del plan                                           # released behind the propagate on stream a
        g = ds.batch(ids)                                  # stream b takes the block: must wait for a's event inside the library
        y2 = gm.propagate(gm.copy_xj, g, "+", xj=g.x)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


# ---- the chain's wave jobs packed on the device -------------------------------------------------------------------------------------------
def check_packing(tab, hdr, sizes):
    """tab [cap][64]: every row of the batch exactly once, member graphs whole and contiguous, jobs filled from slot 0 without holes"""
    njobs = int(hdr[0])
    assert int(hdr[2]) == 0
    seg = np.concatenate([[0], np.cumsum(sizes)])
    N = int(seg[-1])
    used = tab[:njobs]
    assert (tab[njobs:] < 0).all()
    seen = np.zeros(N, np.int64)
    graph_of = np.repeat(np.arange(len(sizes)), sizes)
    tiles = 0
    for j in range(njobs):
        row = used[j]
        n = int((row >= 0).sum())
        assert n > 0 and (row[:n] >= 0).all() and (row[n:] == -1).all(), f"job {j} has a hole"
        np.add.at(seen, row[:n], 1)
        # whole member graphs, each contiguous and in node order
        gs = graph_of[row[:n]]
        at = 0
        while at < n:
            g = gs[at]
            sz = sizes[g]
            assert at + sz <= n and np.array_equal(row[at:at + sz], np.arange(seg[g], seg[g + 1])), f"job {j}: graph {g} is not whole"
            at += sz
        tiles += 2 if n > 32 else 1
    assert (seen == 1).all()
    assert tiles == int(hdr[1])
    return njobs, tiles


@pytest.mark.parametrize("law", ["config5", "uniform64", "small", "big", "single"])
def test_device_packing_is_a_tight_valid_packing(gm, law):
    import torch
    from gnnmp import layers
    rng = np.random.default_rng(3)
    sizes = {"config5": rng.integers(20, 41, 8192), "uniform64": rng.integers(1, 65, 5000), "small": rng.integers(1, 6, 3000),
             "big": rng.integers(33, 65, 700), "single": np.array([17])}[law].astype(np.int64)
    seg = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)])).cuda()
    G, N = len(sizes), int(sizes.sum())
    dev = layers.ChainJobs(seg, G, (N, int(sizes.max()), False))
    host = layers.ChainJobs(seg, G)
    tab, hdr = dev.export()
    njobs, tiles = check_packing(tab.cpu().numpy(), hdr.cpu().numpy(), sizes)
    assert njobs == host.njobs == dev.njobs, "device packing is not as tight as the host's best fit decreasing"
    assert abs(dev.fill - host.fill) < 2e-3
    tabh, hdrh = host.export()
    assert check_packing(tabh.cpu().numpy(), np.array([host.njobs, tiles, 0, 0, 0, 0, 0, 0]), sizes)[0] == njobs


def build_model(gm):
    return gm.GNNChain(gm.GraphConv((16, 128), "relu", seed=21), gm.GraphConv((128, 128), "relu", seed=22),
                       gm.GlobalPool("mean"), gm.Dense((128, 2), seed=23))


def test_chain_on_device_packed_jobs_is_bit_identical(gm):
    import torch
    from gnnmp import synth
    members = synth.batched_graphs(G=1024, seed=4)
    rng = np.random.default_rng(2)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    ds = gm.GraphDataset.from_members(members, xs)
    model = build_model(gm)
    ids = rng.permutation(len(members))[:700]
    g = ds.batch(ids)                                   # plan by select, jobs packed on the device
    y = model(g, g.x)
    assert g._cache["chain_jobs"]._packed
    ref = gm.batch_arrays([members[i] for i in ids], [xs[i] for i in ids])      # sorted plan, host packing
    yr = model(ref, ref.x)
    assert not ref._cache["chain_jobs"]._packed
    assert torch.equal(y, yr)
    assert torch.equal(model(g, g.x), y), "not run-to-run identical"


def test_contradicted_announcement_poisons_the_handle(gm):
    """the caller says `largest member 40 nodes`, the device finds one of 70: no job may run on rows the table does not hold — the chain's
    logits are NaN (loud) and the handle's info call fails"""
    import torch
    from gnnmp import _lib as L, layers
    sizes = np.array([30, 70, 20, 25], np.int64)
    seg = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)])).cuda()
    jobs = layers.ChainJobs(seg, 4, (int(sizes.sum()), 40, False))
    _, hdr = jobs.export()
    assert int(hdr[2]) == 1 and int(hdr[0]) == 0 and int(hdr[3]) == 70
    with pytest.raises(L.GnnmpError):
        jobs.njobs
    # through the chain
    rng = np.random.default_rng(1)
    members = [(np.zeros(0, np.int64), np.zeros(0, np.int64), int(n)) for n in sizes]
    xs = [rng.standard_normal((int(n), 16), dtype=np.float32) for n in sizes]
    g = gm.batch_arrays(members, xs)
    g._cache["member_stats"] = (int(sizes.sum()), 40, False)       # the lie
    y = build_model(gm)(g, g.x)
    assert torch.isnan(y).all()


def test_dataloader_epoch_equals_collated_batches(gm):
    """DataLoader(data; batchsize, shuffle = true, collate = true) over the resident dataset: every step's logits equal those of the batch
    collated the reference's way (batch_arrays = MLUtils.batch + upload + plan_create + host packing)"""
    import torch
    from gnnmp import synth
    members = synth.batched_graphs(G=600, seed=8)
    rng = np.random.default_rng(6)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    y_all = torch.arange(len(members), device="cuda", dtype=torch.float32)
    ds = gm.GraphDataset.from_members(members, xs, targets=y_all)
    model = build_model(gm)
    loader = gm.DataLoader(ds, batchsize=256, shuffle=True, seed=123)
    perm = np.random.default_rng(123).permutation(len(members))
    seen = 0
    for b, (g, yb) in enumerate(loader):
        ids = perm[b * 256:(b + 1) * 256]
        assert torch.equal(yb, y_all[torch.from_numpy(ids).cuda()])
        ref = gm.batch_arrays([members[i] for i in ids], [xs[i] for i in ids])
        assert torch.equal(model(g, g.x), model(ref, ref.x)), b
        seen += len(ids)
    assert seen == len(members) and len(loader) == 3


def test_prefetching_dataloader_yields_the_same_batches(gm):
    """DataLoader(prefetch = True, prepare = chain_prepare): batch k + 1 (plan, features, wave jobs) is prepared on a side stream beside
    batch k's step, the next epoch's first batch behind this epoch's last — the sequence of batches, their targets and the model's logits
    are those of the plain loader, epoch after epoch, also when the consumer is slow or fast (the hand-over is an event, not luck)"""
    import torch
    from gnnmp import synth
    from gnnmp.layers import chain_prepare
    members = synth.batched_graphs(G=700, seed=9)
    rng = np.random.default_rng(7)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    y_all = torch.arange(len(members), device="cuda", dtype=torch.float32)
    ds = gm.GraphDataset.from_members(members, xs, targets=y_all)
    model = build_model(gm)
    plain = gm.DataLoader(ds, batchsize=256, shuffle=True, seed=321)
    pre = gm.DataLoader(ds, batchsize=256, shuffle=True, seed=321, prefetch=True, prepare=chain_prepare)
    big = torch.empty(64 << 20, device="cuda")
    for epoch in range(4):
        want = [(model(g, g.x).clone(), yb.clone(), g.num_nodes) for g, yb in plain]
        got = []
        for b, (g, yb) in enumerate(pre):
            assert g._cache.get("chain_jobs") is not None, "prepare ran on the batch"
            if (epoch + b) % 2:          # a consumer that keeps its stream busy before it gets to the batch
                big.fill_(float(b))
            got.append((model(g, g.x).clone(), yb.clone(), g.num_nodes))
        assert len(got) == len(want) == 3
        for (ya, ta, na), (yb_, tb, nb_) in zip(got, want):
            assert na == nb_ and torch.equal(ta, tb) and torch.equal(ya, yb_), epoch
    torch.cuda.synchronize()
