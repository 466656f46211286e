"""graph_chain_kernel (csrc/graph_chain.hip): GNNChain(GraphConv..., GlobalPool, Dense) on a batched graph in ONE launch — BASELINE.json
config 5 (examples/graph_classification_tudataset.jl:79-82).  Checked against the CPU oracle's layer-by-layer composition
(graph_conv GNNlib/src/layers/conv.jl:102-108, global_pool layers/pool.jl:3-5) within north_star's 1e-5, against the library's own
layer-by-layer path (knob 18 = -1), at the FULL config-5 size G = 8192, and on the shapes that leave the fast lanes of the kernel:
member graphs of 1..70 nodes, isolated nodes, hubs with more than four in-neighbours, mean aggregation, + pooling, one and three
layers, Dout < 128, Din not a multiple of 16, non-finite features."""
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


def random_members(G, rng, nmin=1, nmax=70, hubs=True):
    """member graphs with uneven sizes and degrees: (s, t, n) 1-based local numbering, directed edges, some isolated nodes, some hubs"""
    out = []
    for _ in range(G):
        n = int(rng.integers(nmin, nmax + 1))
        m = int(rng.integers(0, 5 * n + 1))
        s = rng.integers(0, n, m)
        t = rng.integers(0, n, m)
        if hubs and n > 8 and rng.random() < 0.3:              # one node collects 5..n in-edges
            k = int(rng.integers(5, n + 1))
            s = np.concatenate([s, rng.integers(0, n, k)])
            t = np.concatenate([t, np.full(k, int(rng.integers(0, n)))])
        out.append((s.astype(np.int64) + 1, t.astype(np.int64) + 1, n))
    return out


def oracle_chain(oracle, members, xs, convs, pool_aggr, head):
    s, t, gi, n = oracle.batch(members)
    h = np.concatenate(xs)
    for c in convs:
        h = oracle.graph_conv(s, t, n, h, c.weight1.cpu().numpy(), c.weight2.cpu().numpy(),
                              None if c.bias is None else c.bias.cpu().numpy(), c.sigma, c.aggr)
    p = oracle.global_pool(pool_aggr, gi, h, len(members))
    ref = oracle.matmul(head.weight.cpu().numpy(), p)
    if head.bias is not None:
        ref = ref + head.bias.cpu().numpy()[None, :]
    return ref


def build(gm, dims, nout, aggr, pool, sigma="relu", bias=True, seed=21):
    import torch
    convs = [gm.GraphConv((dims[k], dims[k + 1]), sigma, aggr=aggr, bias=bias, seed=seed + 2 * k) for k in range(len(dims) - 1)]
    head = gm.Dense((dims[-1], nout), seed=seed + 50)
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    for c in convs:                                            # non-zero biases (the constructors start them at zero)
        if c.bias is not None:
            c.bias = (torch.rand(c.bias.shape, generator=g) - 0.5).cuda()
    head.bias = (torch.rand(head.bias.shape, generator=g) - 0.5).cuda()
    return gm.GNNChain(*convs, gm.GlobalPool(pool), head)


def run_both(gm, model, g):
    from gnnmp import layers
    assert layers._chain_pattern(model.layers) is not None
    y = model(g, g.x)
    before = gm.knob(18)
    gm.tune(18, -1)
    try:
        y_layers = model(g, g.x)
    finally:
        gm.tune(18, before)
    return y, y_layers


def close(y, ref, tag, k=1.0):
    y = np.asarray(y, np.float64); ref = np.asarray(ref, np.float64)
    assert y.shape == ref.shape, tag
    assert np.linalg.norm(y - ref) <= k * 1e-5 * np.linalg.norm(ref) + 1e-30, (tag, np.linalg.norm(y - ref) / np.linalg.norm(ref))
    assert np.abs(y - ref).max() <= k * 1e-5 * np.abs(ref).max() + 1e-30, (tag, np.abs(y - ref).max() / np.abs(ref).max())


def test_fused_kernel_is_taken(gm):
    """the chain really goes through gnnmp_graphconv_chain_f32 (one launch), not through the layer-by-layer fallback"""
    import torch
    from gnnmp import synth
    members = synth.batched_graphs(G=64, seed=5)
    xs = [np.random.default_rng(1).standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = build(gm, (16, 128, 128), 2, "+", "mean")
    calls = []
    from gnnmp import _lib as L
    lib = L.load()
    real = lib.gnnmp_graphconv_chain_f32

    class Spy:
        def __call__(self, *a):
            rc = real(*a)
            calls.append(rc)
            return rc
    try:
        lib.gnnmp_graphconv_chain_f32 = Spy()
        model(g, g.x)
    finally:
        lib.gnnmp_graphconv_chain_f32 = real
    assert calls == [0]


@pytest.mark.parametrize("G", [64, 512])
def test_config5_shape_vs_oracle_and_layers(gm, oracle, G):
    from gnnmp import synth
    members = synth.batched_graphs(G=G, seed=9)
    rng = np.random.default_rng(1)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = build(gm, (16, 128, 128), 2, "+", "mean")
    y, yl = run_both(gm, model, g)
    ref = oracle_chain(oracle, members, xs, model.layers[:2], "mean", model.layers[-1])
    close(y.cpu().numpy(), ref, "fused vs oracle")
    close(yl.cpu().numpy(), ref, "layers vs oracle")
    import torch
    assert torch.equal(model(g, g.x), y), "not run-to-run identical"


def test_config5_full_size_vs_oracle(gm, oracle):
    """BASELINE.json config 5 as bench.py runs it: G = 8192, N ~ 245 k, GraphConv(16=>128,relu), GraphConv(128=>128,relu),
    GlobalPool(mean), Dense(128=>2) — every logit against the CPU oracle"""
    from gnnmp import synth
    G = 8192
    members = synth.batched_graphs(G=G)
    rng = np.random.default_rng(4)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = gm.GNNChain(gm.GraphConv((16, 128), "relu", seed=21), gm.GraphConv((128, 128), "relu", seed=22),
                        gm.GlobalPool("mean"), gm.Dense((128, 2), seed=23))
    y, yl = run_both(gm, model, g)
    ref = oracle_chain(oracle, members, xs, model.layers[:2], "mean", model.layers[-1])
    close(y.cpu().numpy(), ref, "fused vs oracle, G = 8192")
    close(yl.cpu().numpy(), ref, "layers vs oracle, G = 8192")


CASES = [  # dims, nout, aggr, pool, sigma, bias
    ((16, 128, 128), 2, "mean", "+", "relu", True),
    ((16, 128), 2, "+", "mean", "relu", True),                 # one layer: it is also the last
    ((16, 64, 128, 32), 8, "+", "mean", "relu", True),         # three layers, Dout < 128, nout = 8
    ((20, 36, 128), 1, "mean", "mean", None, False),           # Din % 16 != 0: root and aggregate passes; no σ, no bias
    ((128, 128, 128), 3, "+", "+", "relu", True),              # first layer already needs two passes
    ((4, 8), 2, "+", "mean", "relu", True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[0])) + f"_{c[2]}_{c[3]}")
@pytest.mark.parametrize("seed", [0, 1])
def test_irregular_batches(gm, oracle, case, seed):
    dims, nout, aggr, pool, sigma, bias = case
    rng = np.random.default_rng(100 + seed)
    members = random_members(150, rng)
    xs = [rng.standard_normal((n, dims[0]), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = build(gm, dims, nout, aggr, pool, sigma, bias, seed=31 + seed)
    y, yl = run_both(gm, model, g)
    ref = oracle_chain(oracle, members, xs, model.layers[:-2], pool, model.layers[-1])
    close(y.cpu().numpy(), ref, f"fused vs oracle {case}", k=1.0)
    close(y.cpu().numpy(), yl.cpu().numpy(), f"fused vs layers {case}", k=1.0)


def test_narrow_hidden_layer_is_refused_and_still_right(gm, oracle):
    """a STORED hidden layer of 4 columns is outside the chain kernel's envelope (its unconditional stores re-store a piece at
    column 4 h, i.e. into the next row when the row is 4 wide): the C entry refuses it (GNNMP_EUNSUPPORTED), the host mirror does not
    offer it, and GNNChain still gives the oracle's logits through the layer path"""
    import ctypes
    import torch
    from gnnmp import _lib as L, layers
    rng = np.random.default_rng(77)
    members = random_members(150, rng)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = build(gm, (16, 4, 8), 2, "+", "mean", seed=41)
    assert layers._chain_pattern(model.layers) is None
    y = model(g, g.x)
    ref = oracle_chain(oracle, members, xs, model.layers[:-2], "mean", model.layers[-1])
    close(y.cpu().numpy(), ref, "(16, 4, 8) through the layer path")
    # the C entry itself
    lib = L.load()
    convs, head = model.layers[:2], model.layers[-1]
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    dims = (i64 * 3)(16, 4, 8)
    wr = (vp * 2)(*[c.weight1.data_ptr() for c in convs])
    wa = (vp * 2)(*[c.weight2.data_ptr() for c in convs])
    bs = (vp * 2)(*[c.bias.data_ptr() for c in convs])
    act = (ctypes.c_int * 2)(L.ACT_RELU, L.ACT_RELU)
    N, G = g.num_nodes, g.num_graphs
    sp = torch.empty(G + 1, dtype=torch.int64, device="cuda")
    L.check(lib.gnnmp_segment_bounds(L.ptr(g.graph_indicator), 8, 1, N, G, L.ptr(sp), L.stream_ptr()))
    need = lib.gnnmp_graphconv_chain_scratch_floats(N, 2, dims, 2)
    scratch = torch.empty(max(need, 16), dtype=torch.float32, device="cuda")
    out = torch.empty((G, 2), dtype=torch.float32, device="cuda")
    rc = lib.gnnmp_graphconv_chain_f32(g.plan(False).handle, None, L.ptr(sp), G, L.ptr(g.x), 2, dims, wr, wa, bs, act, 0, L.SUM, L.MEAN,
                                       L.ptr(head.weight), L.ptr(head.bias), 2, L.ptr(scratch), L.ptr(out), L.stream_ptr())
    assert rc == L.EUNSUPPORTED
    # a LAST layer of 4 columns is fine (it is never stored): (16, 8, 4)
    model2 = build(gm, (16, 8, 4), 2, "+", "mean", seed=43)
    y2, yl2 = run_both(gm, model2, g)
    ref2 = oracle_chain(oracle, members, xs, model2.layers[:-2], "mean", model2.layers[-1])
    close(y2.cpu().numpy(), ref2, "(16, 8, 4) fused")


@pytest.mark.parametrize("aggr,pool", [("+", "mean"), ("mean", "+")])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_wave_job_kernel_on_irregular_batches(gm, oracle, aggr, pool, seed):
    """csrc/graph_chain2.hip (two layers 16 => 128 => 128, member graphs <= 64 nodes: a pair of waves per job, no layer output in memory) on
    uneven batches: graphs of 1..64 nodes, isolated nodes, hubs with up to n in-neighbours (the tail loops of both gathers), jobs of
    one and two tiles, empty slots — against the oracle, the general kernel (knob 18 = 1) and the layer-by-layer path"""
    import torch
    from gnnmp import layers
    rng = np.random.default_rng(500 + seed)
    members = random_members(200, rng, nmin=1, nmax=64)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = build(gm, (16, 128, 128), 2 + 3 * seed, aggr, pool, seed=61 + seed)
    y, yl = run_both(gm, model, g)
    jobs = g._cache["chain_jobs"]
    assert jobs.njobs > 0 and jobs.max_graph <= 64 and 0.5 < jobs.fill <= 1.0
    before = gm.knob(18)
    gm.tune(18, 1)
    try:
        y_general = model(g, g.x)
    finally:
        gm.tune(18, before)
    ref = oracle_chain(oracle, members, xs, model.layers[:2], pool, model.layers[-1])
    close(y.cpu().numpy(), ref, "wave-job kernel vs oracle")
    close(y_general.cpu().numpy(), ref, "general kernel vs oracle")
    close(y.cpu().numpy(), yl.cpu().numpy(), "wave-job kernel vs layers")
    for _ in range(3):
        assert torch.equal(model(g, g.x), y), "not run-to-run identical (no atomic on the path: the pooling kernel adds the slabs' halves in a fixed order)"


def ring_graph(n, rng, extra=0):
    """a directed ring over n nodes (every node one in-edge, n = 1: a single isolated node) plus `extra` random edges"""
    if n == 1:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), 1
    s = np.arange(n); t = (s + 1) % n
    if extra:
        s = np.concatenate([s, rng.integers(0, n, extra)]); t = np.concatenate([t, rng.integers(0, n, extra)])
    return s.astype(np.int64) + 1, t.astype(np.int64) + 1, n


@pytest.mark.parametrize("sizes", [[40], [64], [32], [33, 31], [1], [64, 64, 1], [32, 32, 32], [2, 63, 17, 47, 64, 1, 1, 30]],
                         ids=lambda v: "-".join(map(str, v)))
@pytest.mark.parametrize("nout,sigma", [(2, "relu"), (8, None), (1, "relu")])
def test_wave_pair_kernel_edge_sizes(gm, oracle, sizes, nout, sigma):
    """the job shapes at the edges of the pair kernel: a batch of one graph (fewer than 32 rows goes to the layer path: the fused
    kernels need N >= 32), a graph that fills a job exactly, one-tile jobs (the odd wave of the pair idles), a graph that straddles the
    two tiles by one row, single-node graphs, every head width (1, 2: the vector store; 8: the wide one), with and without σ"""
    import torch
    rng = np.random.default_rng(sum(sizes) * 10 + nout)
    members = [ring_graph(n, rng, extra=2 * n if n > 4 else 0) for n in sizes]
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = build(gm, (16, 128, 128), nout, "+", "mean", sigma)
    y, yl = run_both(gm, model, g)
    ref = oracle_chain(oracle, members, xs, model.layers[:2], "mean", model.layers[-1])
    close(y.cpu().numpy(), ref, f"fused vs oracle {sizes}")
    close(yl.cpu().numpy(), ref, f"layers vs oracle {sizes}")
    assert torch.equal(model(g, g.x), y)


def test_job_packing(gm):
    """gnnmp_chain_jobs_create: every row in exactly one slot, member graphs whole and in node order inside a job, at most 64 rows a
    job; a batch with a larger member graph gets no jobs (the general kernel runs)"""
    import ctypes, torch
    from gnnmp import _lib as L, synth
    members = synth.batched_graphs(G=1000, seed=7)
    g = gm.batch_arrays(members, [np.zeros((n, 16), np.float32) for _, _, n in members])
    model = build(gm, (16, 128, 128), 2, "+", "mean")
    model(g, g.x)
    jobs = g._cache["chain_jobs"]
    assert jobs.fill >= 0.9, jobs.fill                       # n ~ U{20..40}: best fit decreasing fills the 64-row jobs to ~94 %
    rng = np.random.default_rng(1)
    big = random_members(30, rng, nmin=60, nmax=90)
    g2 = gm.batch_arrays(big, [np.zeros((n, 16), np.float32) for _, _, n in big])
    model(g2, g2.x)
    assert g2._cache["chain_jobs"].njobs == 0 and g2._cache["chain_jobs"].max_graph > 64


def test_sharded_equals_unsharded_bit_for_bit(gm):
    """member graphs are independent units and a row's arithmetic does not depend on where its tile falls: any regrouping of the
    batch (gnnmp.parallel shards by graph) reproduces the same logits bit for bit"""
    import torch
    from gnnmp import synth
    from gnnmp.parallel import shard_by_size
    G = 300
    members = synth.batched_graphs(G=G, seed=2)
    rng = np.random.default_rng(3)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    model = build(gm, (16, 128, 128), 2, "+", "mean")
    g = gm.batch_arrays(members, xs)
    y = model(g, g.x)
    shards = shard_by_size([m[2] for m in members], 3)
    merged = torch.empty_like(y)
    for r in range(3):
        gr = gm.batch_arrays([members[i] for i in shards[r]], [xs[i] for i in shards[r]])
        merged[torch.as_tensor(shards[r], device="cuda")] = model(gr, gr.x)
    assert torch.equal(merged, y)


def test_non_finite_features_take_the_exact_path(gm):
    """an Inf / NaN feature makes the accumulators of its tile NaN on the split-bf16 core; the kernel re-scans the pass and redoes
    such tiles with fp32 fma loops: the affected graphs' logits are non-finite exactly where the layer-by-layer path's are, every
    other graph is untouched"""
    import torch
    from gnnmp import synth
    G = 200
    members = synth.batched_graphs(G=G, seed=4)
    rng = np.random.default_rng(5)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    xs[17][3, 2] = np.inf
    xs[101][0, 0] = np.nan
    xs[150][5] = -np.inf
    g = gm.batch_arrays(members, xs)
    model = build(gm, (16, 128, 128), 2, "+", "mean")
    y, yl = run_both(gm, model, g)
    bad = (~torch.isfinite(y).all(1)).nonzero().flatten().tolist()
    assert bad == (~torch.isfinite(yl).all(1)).nonzero().flatten().tolist() == [17, 101, 150]
    good = torch.isfinite(y).all(1)
    close(y[good].cpu().numpy(), yl[good].cpu().numpy(), "finite graphs")
    assert torch.equal(torch.isnan(y), torch.isnan(yl))


@pytest.mark.parametrize("dims", [(16, 128, 128), (16, 64, 128)])
def test_julia_weight_layout(gm, dims):
    """w_layout = 1: Julia's (out, in) column-major weight matrices as stored (C row-major [in][out]) — what the Julia binding passes —
    give the same logits, bit for bit, as the C row-major [out][in] matrices (both fused kernels)"""
    import ctypes
    import torch
    from gnnmp import _lib as L, synth
    from gnnmp.msgpass import aggr_code
    lib = L.load()
    members = synth.batched_graphs(G=200, seed=11)
    xs = [np.random.default_rng(2).standard_normal((n, dims[0]), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    model = build(gm, dims, 3, "+", "mean")
    y = model(g, g.x)
    convs, head = model.layers[:-2], model.layers[-1]
    nl = len(convs)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    wt = [(c.weight1.t().contiguous(), c.weight2.t().contiguous()) for c in convs]      # [in][out] = Julia (out, in) as stored
    ht = head.weight.t().contiguous()
    cdims = (i64 * (nl + 1))(*dims)
    wr = (vp * nl)(*[w[0].data_ptr() for w in wt])
    wa = (vp * nl)(*[w[1].data_ptr() for w in wt])
    bs = (vp * nl)(*[c.bias.data_ptr() for c in convs])
    act = (ctypes.c_int * nl)(*[1] * nl)
    out = torch.empty_like(y)
    scratch = torch.empty(lib.gnnmp_graphconv_chain_scratch_floats(g.num_nodes, nl, cdims, 3), device="cuda")
    L.check(lib.gnnmp_graphconv_chain_f32(g.plan(False).handle, g._cache["chain_jobs"].handle, L.ptr(g._cache["node_ptr"]), g.num_graphs,
                                          L.ptr(g.x), nl, cdims, wr, wa, bs, act, 1, aggr_code("+"), aggr_code("mean"), L.ptr(ht),
                                          L.ptr(head.bias), 3, L.ptr(scratch), L.ptr(out), L.stream_ptr()))
    assert torch.equal(out, y)


@pytest.mark.parametrize("din,dims", [(7, (128, 128)), (3, (128, 128)), (16, (128, 128)), (7, (64, 32)), (21, (128, 128)), (1, (128,))])
def test_any_input_width_is_zero_padded_into_the_fused_kernels(gm, oracle, din, dims):
    """TUDataset node features are one-hot labels (MUTAG 7, PROTEINS 3: examples/graph_classification_tudataset.jl:74-82 builds
    GraphConv(nin => 128)): the host pads the features and the first layer's weight columns with zeros — exact — to 16 (the wave-pair
    kernel's shape) or to the next multiple of 4 (the general kernel); the fused entry point is taken and the logits are the oracle's"""
    import torch
    from gnnmp import _lib as L
    rng = np.random.default_rng(din * 31 + len(dims))
    members = random_members(300, rng, nmin=10, nmax=28, hubs=False)
    xs = []
    for _, _, n in members:                                    # one-hot node labels
        lab = rng.integers(0, din, n)
        xs.append(np.eye(din, dtype=np.float32)[lab])
    g = gm.batch_arrays(members, xs)
    model = build(gm, (din,) + dims, 2, "+", "mean")
    lib = L.load()
    real, calls = lib.gnnmp_graphconv_chain_f32, []

    class Spy:
        def __call__(self, *a):
            rc = real(*a)
            calls.append(rc)
            return rc
    try:
        lib.gnnmp_graphconv_chain_f32 = Spy()
        y = model(g, g.x)
    finally:
        lib.gnnmp_graphconv_chain_f32 = real
    assert calls == [0], "the chain did not run on the fused kernel"
    if dims == (128, 128) and din <= 16:
        assert g._cache["chain_jobs"].njobs > 0                # the wave-pair kernel's envelope
    ref = oracle_chain(oracle, members, xs, model.layers[:-2], "mean", model.layers[-1])
    close(y.cpu().numpy(), ref, f"din={din} dims={dims}")
    before = gm.knob(18)
    gm.tune(18, -1)
    try:
        close(model(g, g.x).cpu().numpy(), ref, f"layers din={din}")
    finally:
        gm.tune(18, before)
    # replacing a weight in place (an optimiser step) invalidates the cached padded copy
    with torch.no_grad():
        model.layers[0].weight1.mul_(0.5)
    ref2 = oracle_chain(oracle, members, xs, model.layers[:-2], "mean", model.layers[-1])
    close(model(g, g.x).cpu().numpy(), ref2, "after an in-place weight update")


def test_outside_the_envelope_falls_back(gm, oracle):
    """max aggregation, a wide layer, a non-batched graph: gnnmp_graphconv_chain_f32 says GNNMP_EUNSUPPORTED (or the host does not
    even ask) and the chain runs layer by layer with the same result contract"""
    rng = np.random.default_rng(8)
    members = random_members(40, rng, nmin=5, nmax=30, hubs=False)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    g = gm.batch_arrays(members, xs)
    for dims, aggr in (((16, 128), "max"), ((16, 256, 128), "+")):
        model = build(gm, dims, 2, aggr, "mean")
        y = model(g, g.x)
        ref = oracle_chain(oracle, members, xs, model.layers[:-2], "mean", model.layers[-1])
        finite = np.isfinite(ref).all(1)
        close(y.cpu().numpy()[finite], ref[finite], f"fallback {dims} {aggr}")
