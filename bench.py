#!/usr/bin/env python
"""bench.py — forward edges/sec of GCNConv + GATConv on an ogbn-products-shaped synthetic graph (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: one GCNConv(100=>100, relu) forward plus one
GATConv(100=>16, heads=8, relu) forward on the products-shaped graph (N=2 449 029, E=61 859 140, +N self loops each),
features already resident in HBM, plans (dst-sorted CSR) built before the timed region and reported separately.
value = edges traversed per second, whole job = n_gpus * 2 * E' / step time (ranks are independent replicas with
different feature batches: weak scaling, no data-path collective — SURVEY.md §8e).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload products|arxiv|batched|rowpart] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself under torch.distributed.run with N
ranks on 127.0.0.1 (one process per GPU, RCCL); it exits non-zero — never degrades to one rank — if the node has fewer GPUs.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — the dominant kernel's algorithmic bytes / its average duration measured with HIP events on the launch
                 stream inside the timed region, against the 8 TB/s HBM peak (MI355X_MICROARCH.md)
  cpu_baseline — the CPU oracle (a port of the reference's algorithm; Julia is not installed) timed on the host, rank 0,
                 N=1 only, on a bounded sample: the same two layers on a 1/32-scale products-shaped graph, plus the GCN layer
                 once on the full products-shaped graph (the same workload as the GPU side).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_ACHIEVABLE_GBS = 6290.0  # ... that measured ceiling: roofline.frac_of_achievable ("fraction of HBM peak flatters")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def alg_bytes_gcn_propagate(N, Ep, D):
    """fused GCN propagate kernel (csr_rows_kernel): per edge one source row + its col id (SURVEY.md §8d 'CSR SpMM':
    4D + 4); per node the output row, rowptr (8 B) and the two normalisation scalars ('GCN layer (fused norm)': + 8 N).
    The slot-ordered source coefficient the kernel actually streams (4 B/edge more) is NOT counted: conservative."""
    return Ep * (4 * D + 4) + 8 * (N + 1) + 4 * N * D + 8 * N


def alg_bytes_gat_aggregate(N, Ep, H, C):
    """one-pass GAT kernel (gat_fused_rows_kernel): per edge one source row (4HC) and its col id (4); per node the
    output row (4HC), the node's own Wx row for the target half of the logit (4HC) and rowptr (8).  Below SURVEY.md
    §8d's 'GAT fused' figure (which budgets 8H + 16 B/edge of scores and Int64 indices): conservative."""
    return Ep * (4 * H * C + 4) + N * (8 * H * C + 8)


def compulsory_bytes(N, Ep, D_in, D_out):
    """SURVEY.md §8d's compulsory bound for an aggregation: every input row read once, every output row written once, and
    the index (int32 col per edge + rowptr).  What an ideal cache would leave of the algorithmic bytes."""
    return 4 * N * D_in + 4 * N * D_out + 4 * Ep + 4 * (N + 1)


def mem_available_gib():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def cpu_baseline(products_graph=None, full_gat=False):
    """The reference's algorithm (oracle port), single thread, on the paths the reference takes with CPU arrays: GCNConv through
    the SpMM fast path (COO -> CSC rebuild + CSC sweep per call, msgpass.jl:215-238), GATConv through gather -> message ->
    scatter with materialised (D, E') temporaries.  Sample: both layers of the bench step once on a 1/32-scale products-shaped
    graph (same generator, same degree law, D = 100), and — when the caller passes the full graph — the GCN layer once at
    full size, so that one leg of the ratio is the identical workload."""
    import numpy as np
    from gnnmp import synth
    from oracle import oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    orc.build()
    D, H, C = synth.PRODUCTS["D"], 8, 16
    SCALE = 32
    N = synth.PRODUCTS["N"] // SCALE
    E = (synth.PRODUCTS["E"] // SCALE) & ~1
    s, t = synth.products_like(N=N, E=E, seed=5)
    x = synth.features(N, D, seed=1)
    rng = np.random.default_rng(0)
    W = (rng.standard_normal((D, D)) * 0.1).astype(np.float32)
    b = np.zeros(D, np.float32)
    Wd = (rng.standard_normal((H * C, D)) * 0.1).astype(np.float32)
    ba = np.zeros(H * C, np.float32)
    a = (rng.standard_normal((2 * C, H)) * 0.1).astype(np.float32)
    Ep = len(s) + N

    def run():
        t0 = time.perf_counter()
        orc.gcn_conv(s, t, N, x, W, b, "relu", fast_path=True)
        t1 = time.perf_counter()
        orc.gat_conv(s, t, N, x, Wd, a, ba, "relu", heads=H)
        t2 = time.perf_counter()
        orc.gcn_conv(s, t, N, x, W, b, "relu", fast_path=False)     # the generic path for GCN too (what its GPU ext does)
        t3 = time.perf_counter()
        full = full_a = None
        if products_graph is not None:
            sf, tf, xf = products_graph
            t4 = time.perf_counter()
            orc.gcn_conv(sf, tf, xf.shape[0], xf, W, b, "relu", fast_path=True)
            full = time.perf_counter() - t4
            # --cpu-baseline-full: the GAT leg at full size too.  The reference's path materialises Wxi, Wxj and β
            # as (C, H, E') arrays — 3 x 33 GB here, plus the softmax temporaries — so it runs only on a host with the memory for it.
            if full_gat and mem_available_gib() >= 256.0:
                t5 = time.perf_counter()
                orc.gat_conv(sf, tf, xf.shape[0], xf, Wd, a, ba, "relu", heads=H)
                full_a = time.perf_counter() - t5
        return t1 - t0, t2 - t1, t3 - t2, full, full_a

    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            tg, ta, tgg, tfull, tfull_a = run()
    else:
        tg, ta, tgg, tfull, tfull_a = run()
    out = {
        "value": 2 * Ep / (tg + ta), "unit": "edges/s", "cores": 1, "kind": "port", "cpu_model": host_cpu_model(),
        "host_cores": os.cpu_count(),
        "sample": f"1/{SCALE}-scale products-shaped graph N={N} E'={Ep} D={D}: GCNConv({D}=>{D},relu)+GATConv({D}=>{C},h={H},relu) "
                  f"forward once on the paths the reference takes with CPU arrays: GCN through the SpMM fast path (COO->CSC "
                  f"rebuild + CSC sweep per call) {tg:.2f}s, GAT through gather->message->scatter with materialised (D,E') "
                  f"temporaries {ta:.2f}s"
                  + (f"; plus the GCN layer once on the FULL products-shaped graph: {tfull:.1f}s" if tfull else ""),
        "gcn_edges_per_s": Ep / tg, "gat_edges_per_s": Ep / ta, "gcn_generic_path_edges_per_s": Ep / tgg,
    }
    if tfull:
        Epf = len(products_graph[0]) + products_graph[2].shape[0]
        out["gcn_full_products"] = {"E_prime": Epf, "seconds": tfull, "edges_per_s": Epf / tfull}
        if tfull_a:
            out["gat_full_products"] = {"E_prime": Epf, "seconds": tfull_a, "edges_per_s": Epf / tfull_a}
            out["full_products_value"] = {"edges_per_s": 2 * Epf / (tfull + tfull_a),
                                          "what": "both layers of the step once at FULL size on one core: the identical workload to `value`"}
        elif full_gat:
            out["gat_full_products"] = {"skipped": f"MemAvailable {mem_available_gib():.0f} GiB < 256 GiB (the path materialises ~110 GB of (C,H,E') temporaries)"}
    try:
        # a fair "best CPU" line — OpenMP CSR aggregation on every host core, CSR built once (NOT the
        # reference's algorithm: its CPU propagate is single-threaded)
        from oracle import allcore
        allcore.lib(rebuild=True)                                   # -march=native for THIS host
        rp, col = allcore.build_csr(s, t, N)
        c = orc.inv_sqrt(orc.degree(orc.add_self_loops(s, t, N)[1], N))
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            allcore.spmm(rp, col, x, c)
            best = min(best, time.perf_counter() - t0)
        out["best_cpu_allcore"] = {"what": "OpenMP CSR normalised sum-aggregation (GCN propagate only), best of 5",
                                   "threads": allcore.threads(), "edges_per_s": Ep / best, "ms": best * 1e3}
    except Exception as e:  # pragma: no cover — a missing OpenMP runtime must not sink the bench line
        out["best_cpu_allcore"] = {"error": repr(e)}
    return out


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def julia_reference_baseline():
    """SURVEY.md §8d's preferred CPU baseline: the reference itself (bench/reference_cpu.jl, BenchmarkTools, the pattern of
    GraphNeuralNetworks/perf/perf.jl:25) — used iff a `julia` with the reference's packages is on this host; else None and the oracle port
    above is the baseline (`kind: "port"`)."""
    import shutil
    import subprocess
    jl = shutil.which("julia")
    script = os.path.join(ROOT, "bench", "reference_cpu.jl")
    if jl is None or not os.path.exists(script):
        return None
    try:
        r = subprocess.run([jl, "--project=" + os.path.join(ROOT, "bench"), script, "--scale", "32"], capture_output=True, text=True, timeout=900)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                d = json.loads(line)
                d["kind"] = "reference"
                return d
    except Exception as e:  # pragma: no cover
        return {"error": repr(e)}
    return None


def timed_region(step, steps, warmup, barrier, dist, device="cuda", before=None, after=None):
    """The driver's protocol, in one place for every workload: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by
    barrier() (dist.barrier + device synchronise) on both sides, wall clock, MAX over the ranks.  Returns (seconds, last output).
    `before` / `after` run inside the brackets, right around the timed loop (the kernel event probe).  Shared by the GPU run and by the
    world-2 / world-8 gloo rehearsals of tests/test_parallel_gloo.py (device = "cpu" there)."""
    import torch
    out = None
    for _ in range(warmup):
        step()
    barrier()
    if before is not None:
        before()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if after is not None:
        after()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, out


def batched_setup(rank, world, dist, G=8192, forward_factory=None, device=None):
    """BASELINE.json config 5 on this rank: its shard of the G synthetic graphs batched on the device, the model, and the static part of
    the exchange (gnnmp.parallel.ShardPlan: send / receive buffers and the inverse permutation, built ONCE).  Returns step() — model
    forward + one all-gather of the (G_r, 2) logits + one index_select, device work only — and the problem sizes.
    forward_factory(members, xs) -> callable returning the (len(members), 2) logits: the HIP product by default; the CPU dry-run of
    this very code path (tests/test_parallel_gloo.py, world 2 over gloo) passes the oracle's."""
    import numpy as np
    import torch
    from gnnmp import synth
    from gnnmp.parallel import ShardPlan, shard_by_size
    members = synth.batched_graphs(G=G)
    rng = np.random.default_rng(4)
    xs = [rng.standard_normal((n, 16), dtype=np.float32) for _, _, n in members]
    shards = shard_by_size([m[2] for m in members], world)
    mine = shards[rank]
    if forward_factory is None:
        import gnnmp
        g = gnnmp.batch_arrays([members[i] for i in mine], [xs[i] for i in mine])
        model = gnnmp.GNNChain(gnnmp.GraphConv((16, 128), "relu", seed=21), gnnmp.GraphConv((128, 128), "relu", seed=22),
                               gnnmp.GlobalPool("mean"), gnnmp.Dense((128, 2), seed=23))
        g.plan(False)
        model(g, g.x)
        # per-batch constants — outside the timed steps — measured on a SECOND batch object of the same graphs once the library is warm:
        # the plan of the batched graph and the chain jobs (a training loop pays both once per NEW batch)
        g2 = gnnmp.batch_arrays([members[i] for i in mine], [xs[i] for i in mine])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g2.plan(False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model(g2, g2.x)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        model(g2, g2.x)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        batched_setup.prep_ms = {"plan_ms": (t1 - t0) * 1e3, "chain_jobs_ms": ((t2 - t1) - (t3 - t2)) * 1e3,
                                 "what": "the reference's own way per new batch: MLUtils.batch on the host + upload (not timed here) + "
                                         "gnnmp_plan_create (sort) + gnnmp_chain_jobs_create (host packing)"}
        del g2
        # The loop the reference's example runs (graph_classification_tudataset.jl:70-71, 97-104): a NEW batch every step, here from the
        # device-resident dataset (gnnmp.GraphDataset / DataLoader: plan by gnnmp_plan_select, features by one gather, wave jobs packed
        # on the device by gnnmp_chain_jobs_pack — no sort, no host synchronisation).  Every step is a fresh shuffle of the G members.
        ds = gnnmp.GraphDataset.from_members([members[i] for i in mine], [xs[i] for i in mine])
        loader = gnnmp.DataLoader(ds, batchsize=len(mine), shuffle=True, seed=7)

        def epochs(n, run_model):
            torch.cuda.synchronize(); ta = time.perf_counter()
            for _ in range(n):
                for gb in loader:
                    if run_model:
                        yb = model(gb, gb.x)
                    else:
                        gnnmp.layers.ChainJobs(gb._cache["node_ptr"], gb.num_graphs, gb._cache["member_stats"])
            torch.cuda.synchronize()
            return (time.perf_counter() - ta) / n * 1e3

        epochs(5, True)
        t_new = epochs(50, True)
        t_prep = epochs(50, False)
        # the same loop with the loader's prefetch (round 6): batch k + 1 — plan, features, wave jobs — prepared on a side stream beside
        # batch k's step; same batches, same logits (tests/test_plan_batch.py)
        loader_single = loader
        loader = gnnmp.DataLoader(ds, batchsize=len(mine), shuffle=True, seed=7, prefetch=True, prepare=gnnmp.layers.chain_prepare)
        epochs(5, True)
        t_new_prefetch = epochs(50, True)
        loader = loader_single

        def prep_sync(n):          # one batch at a time, synchronised: the latency of one preparation (nothing to overlap with)
            tot = 0.0
            for _ in range(n):
                torch.cuda.synchronize(); ta = time.perf_counter()
                for gb in loader:
                    jb = gnnmp.layers.ChainJobs(gb._cache["node_ptr"], gb.num_graphs, gb._cache["member_stats"])
                torch.cuda.synchronize(); tot += time.perf_counter() - ta
                del gb, jb
            return tot / n * 1e3
        batched_setup.new_batch = {
            "batched_new_batch_every_step_ms": t_new_prefetch, "batched_new_batch_every_step_ms_single_stream": t_new,
            "prep_only_ms_pipelined": t_prep, "prep_only_ms_synchronised": prep_sync(20),
            "what": "per step: shuffle on the host, ONE upload of the permutation (64 KB), gnnmp_plan_select (2 launches), gnnmp_gather_f32 "
                    "of the features, gnnmp_chain_jobs_pack (1 launch), then the fused chain (2 launches); no host synchronisation in "
                    "the loop (the host prepares step k + 1 while the device runs step k); the first figure with the preparation of "
                    "step k + 1 on a side stream beside step k's kernels (DataLoader(prefetch = True)), the second with everything on one stream"}
        del loader, loader_single, ds
        # The TRAINING step of the same model on the same batch (the reference example is a training script,
        # graph_classification_tudataset.jl:97-104): forward layer by layer with stored activations, every pullback a libgnnmp adjoint
        # kernel (gnnmp/backward.py), torch for the loss and Adam.  There is no fused backward chain: reported, not claimed.
        try:
            import torch.nn.functional as F
            from gnnmp.backward import dense_ad, global_pool_ad, graph_chain_ad, graph_conv_ad
            c1, c2, poolL, head = model.layers
            tparams = [c1.weight1, c1.weight2, c1.bias, c2.weight1, c2.weight2, c2.bias, head.weight, head.bias]
            saved = [p.detach().clone() for p in tparams]
            for p in tparams:
                p.requires_grad_(True)
            opt = torch.optim.Adam(tparams, lr=1e-3, fused=True)
            Yb = torch.from_numpy(np.random.default_rng(8).integers(0, 2, g.num_graphs)).cuda()

            def train_steps(n, chain=True):
                torch.cuda.synchronize(); ta = time.perf_counter()
                for _ in range(n):
                    opt.zero_grad(set_to_none=True)
                    lg = graph_chain_ad(model, g, g.x) if chain else dense_ad(head, global_pool_ad(poolL, g, graph_conv_ad(c2, g, graph_conv_ad(c1, g, g.x))))
                    F.cross_entropy(lg, Yb).backward()
                    opt.step()
                torch.cuda.synchronize()
                return (time.perf_counter() - ta) / n * 1e3
            train_steps(3)
            t_train = train_steps(20)
            train_steps(3, chain=False)
            t_layers = train_steps(20, chain=False)
            with torch.no_grad():
                for p, v in zip(tparams, saved):
                    p.copy_(v)
            for p in tparams:
                p.requires_grad_(False)
                p.grad = None
            batched_setup.train = {"train_step_ms": t_train, "train_step_ms_layer_by_layer_adjoints": t_layers,
                                   "what": "forward (activations stored) + cross-entropy + the chain's scheduled pullback "
                                           "(gnnmp.backward.graph_chain_ad: pool-grad + relu' in one pass, both weight gradients of a "
                                           "GraphConv from one read of dz, the x-pullback's sum and relu' in the row kernel's epilogue) "
                                           "+ Adam on the static batch; bit-identical gradients to the layer-by-layer adjoints"}
        except Exception as e:      # the side line must not take the bench line with it
            batched_setup.train = {"error": repr(e)}
        forward = lambda: model(g, g.x)                      # noqa: E731
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        forward = forward_factory([members[i] for i in mine], [xs[i] for i in mine])
    plan = ShardPlan(shards, rank, world, 2, device, torch.float32, dist)

    def step():
        return plan.gather(forward())

    n_tot = sum(m[2] for m in members)
    e_tot = sum(len(m[0]) for m in members)
    return step, G, n_tot, e_tot


def run_batched(args, rank, world, dist, barrier):
    """BASELINE.json config 5: G = 8192 synthetic graphs x ~30 nodes, GNNChain(GraphConv(16=>128,relu),
    GraphConv(128=>128,relu), GlobalPool(mean), Dense(128=>2)), member graphs sharded by graph across the ranks
    (gnnmp.parallel), one all-gather of the (G_r, 2) logits per step.  Total work is fixed: strong scaling."""
    import torch
    step, G, n_tot, e_tot = batched_setup(rank, world, dist)
    dt, out = timed_region(step, args.steps, args.warmup, barrier, dist)
    assert out.shape == (G, 2)
    result = {
        "metric": "graphs/sec (fwd) batched graph classification, GraphConv x2 + GlobalPool(mean) + Dense",
        "value": G / (dt / args.steps), "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"batched: G={G} graphs, N={n_tot} nodes, E={e_tot} edges, D=16=>128=>128=>2; "
                               f"edges/s = {2 * e_tot / (dt / args.steps):.3e} (two GraphConv layers)",
                   "parallelism": f"graph-parallel x{world}: shard by graph, one all-gather of (G_r,2) logits per step"},
    }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_rowpart(args, rank, world, dist, barrier):
    """One products-shaped graph partitioned by destination rows across the ranks (gnnmp.rowpart, SURVEY.md §8e "next"):
    two GCNConv(100=>100, relu) layers per step, each followed by ONE all-gather of the (N, 100) layer output.  Total work
    is fixed: strong scaling; communication-bound by design (§8e: 122 MB leave every rank per layer at world = 8)."""
    import numpy as np
    import torch
    import gnnmp
    from gnnmp import _lib as L, rowpart as RP, synth
    from gnnmp.graph import Plan
    N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
    s, t = synth.products_like()
    g = gnnmp.add_self_loops(gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N,
                                            _validated=True))
    lib = L.load()
    deg = torch.empty(N, dtype=torch.float32, device="cuda")
    L.check(lib.gnnmp_degree_f32(g.plan(False).handle, None, L.ptr(deg), L.stream_ptr()))
    c = torch.empty_like(deg)
    L.check(lib.gnnmp_inv_sqrt_f32(L.ptr(deg), L.ptr(c), N, L.stream_ptr()))
    bounds = RP.partition_rows_by_edges(g.t, N, world)
    lo, hi = bounds[rank]
    sl, tl, keep = RP.local_edges(g.s, g.t, lo, hi)
    plan = Plan(sl, tl, N, hi - lo, 1, False, validate=False)
    cl = c[lo:hi].contiguous()
    del g
    torch.cuda.empty_cache()
    layers = [gnnmp.GCNConv((D, D), "relu", seed=31), gnnmp.GCNConv((D, D), "relu", seed=32)]
    x = torch.from_numpy(synth.features(N, D, seed=1)).cuda()
    agg = torch.empty((hi - lo, D), dtype=torch.float32, device="cuda")

    def local_layer(l):
        def f(h):
            # one kernel when the rank's aggregate cannot stay in the Infinity Cache (the library decides), else propagate + dense
            y = gnnmp.fused_conv(plan, L.SUM, h, l.weight, l.bias, "relu", scale_src=c, scale_dst=cl)
            if y is not None:
                return y
            L.check(lib.gnnmp_propagate_f32(plan.handle, L.COPY_XJ, L.SUM, L.ptr(h), None, L.ptr(c), L.ptr(cl), L.ptr(agg), D,
                                            L.stream_ptr()))
            return gnnmp.dense(agg, l.weight, l.bias, "relu")
        return f

    fs = [local_layer(l) for l in layers]

    def step():
        return RP.row_parallel_forward(fs, x, bounds, rank, world, dist)

    dt, out = timed_region(step, args.steps, args.warmup, barrier, dist)
    assert out.shape == (N, D)
    Ep = E + N
    result = {
        "metric": "edges/sec (fwd) 2 x GCNConv, ogbn-products-shape, one graph row-partitioned across the GPUs",
        "value": 2 * Ep / (dt / args.steps), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"products-shape N={N} E'={Ep} D={D}: 2 x GCNConv({D}=>{D},relu); rank rows {lo}..{hi} "
                               f"({int(keep.numel())} edges on rank {rank})",
                   "parallelism": f"row-partition x{world}: per-rank plan over its destination rows, one all-gather of the "
                                  f"(N,{D}) output per layer ({4 * N * D / max(world, 1) / 1e6:.0f} MB per rank)"},
    }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: become N ranks under torch.distributed.run on 127.0.0.1."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); refusing to run fewer ranks")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] self-launch:", " ".join(cmd))
    raise SystemExit(subprocess.call(cmd, env=env))


def traffic_from_profiles(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (bench.py cannot collect counters itself): value plus
    where it came from, so that a stale number is visible in the line."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        pt = doc.get(workload, {}).get(kernel)
        if pt:
            return pt["hbm_read_bytes"] + pt["hbm_write_bytes"], {
                "file": "profiles/pmc_traffic.json", "passes": pt.get("source"), "round": pt.get("round", doc.get("_round")),
                "measured_at_commit": pt.get("commit", doc.get("_commit")),
                "note": "constant from the committed rocprofv3 --pmc passes of this same command, not measured by this run"}
    except OSError:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="products", choices=["products", "arxiv", "batched", "rowpart"],
                    help="products (default, the BASELINE.json metric) | arxiv | batched (config 5: 8192 graphs sharded "
                         "by graph across the ranks, one RCCL all-gather of logits per step; strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="also time the GATConv leg of the CPU baseline once at FULL products size (~40 s of one core and ~110 GB of host "
                         "memory for the reference path's (C,H,E') temporaries; skipped below 256 GiB available)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (other configs, 8d protocol)")
    ap.add_argument("--no-placement", action="store_true", help="fresh output allocations per layer call instead of gnnmp.placement's persistent, placement-tuned buffers")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if args.gpus > 1 and not launched:
        self_launch(args)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local_rank)
    dist = None
    if launched:                               # launched by torch.distributed.run (also for world == 1)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import gnnmp
    from gnnmp import _lib as L
    from gnnmp import synth
    from gnnmp.layers import gcn_norm_cache
    gnnmp.load()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "batched":
        return run_batched(args, rank, world, dist, barrier)
    if args.workload == "rowpart":
        return run_rowpart(args, rank, world, dist, barrier)

    # ---- workload -------------------------------------------------------------------------------------------
    if args.workload == "products":
        N, E, D = synth.PRODUCTS["N"], synth.PRODUCTS["E"], synth.PRODUCTS["D"]
        t0 = time.perf_counter()
        s, t = synth.products_like()
        log(f"[rank {rank}] products-shaped graph generated in {time.perf_counter() - t0:.1f}s")
    else:
        N, E, D = synth.ARXIV["N"], synth.ARXIV["E"], synth.ARXIV["D"]
        s, t = synth.arxiv_like()
    H, C = 8, 16
    Ep = E + N
    x_host = synth.features(N, D, seed=1 + rank)
    x = torch.from_numpy(x_host).cuda()
    sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    keep_host = rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "products"
    if not keep_host:
        del s, t, x_host
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = gnnmp.GNNGraph(sd, td, num_nodes=N, _validated=True)
    plan = g.plan(True)          # both layers add self loops: one plan
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3
    gcn = gnnmp.GCNConv((D, D), "relu", seed=11)
    gat = gnnmp.GATConv((D, C), "relu", heads=H, seed=12)
    # per-graph GCN constants (degree, 1/sqrt(d), slot-ordered coefficients): plan-time work like the CSR itself, done once per
    # graph and reported next to plan_create_ms (the reference recomputes them inside every call)
    t0 = time.perf_counter()
    cvec, c_slot, _ = gcn_norm_cache(g, True, None)
    torch.cuda.synchronize()
    norm_cache_ms = (time.perf_counter() - t0) * 1e3

    # Output placement (gnnmp/placement.py, LABNOTES.md §5 round 4): both gather kernels run 6 % faster or slower depending on where the
    # gathered matrix and the output lie relative to each other in device memory (same binary, same data); the layers keep a persistent
    # output buffer whose placement they found fastest by timing themselves on three candidates.  --no-placement: fresh allocations per
    # call, the mode is then whatever the allocator hands out (the r3 behaviour).
    gcn.place_outputs = gat.place_outputs = not args.no_placement

    x_plain = x                # the features as torch's allocator placed them (extras.value_without_arena)

    def step():
        y1 = gcn(g, x)
        y2 = gat(g, x)
        return y1, y2

    use_arena = False
    if not args.no_placement:          # builds the arena (set-up, like the plan) and moves the input features into it
        # The probing budget is the CALLER's choice (gnnmp_arena_create's max_probe_bytes / GNNMP_ARENA_BUDGET_MS).  hipMalloc hands out
        # physical memory in runs of one placement class that can be tens of GiB long: with the library's default (32 GiB held, 0.3 s) two of
        # this round's four boxes found only two classes — and the step is 4-5 % slower there.  The bench owns the whole 288 GB device, so
        # it lets the search hold up to 112 GiB for up to 2 s (still set-up, reported as arena_create_and_first_step_ms); the environment
        # overrides both.
        if args.workload == "products":      # (the arxiv-shaped buffers are below placement's size floor: the default budget will do)
            os.environ.setdefault("GNNMP_ARENA_PROBE_GIB", "112")
            os.environ.setdefault("GNNMP_ARENA_BUDGET_MS", "2000")
        t0 = time.perf_counter()
        ar0 = gnnmp.placement.arena()
        if ar0 is not None:
            # the caller's side of the recipe: node features allocated from the arena have a KNOWN placement class — an
            # ordinary allocation is a patchwork of physical blocks of any class, and a gathered matrix that shares part of its class with
            # the output gets part of the slow mode (measured: GCN layer 4.86 ms with x from torch's allocator, 4.67 ms with x from the arena)
            xa = ar0.alloc((N, D), 0)
            if xa is not None:
                xa.copy_(x)
                x = xa
        step()
        torch.cuda.synchronize()
        arena_ms = (time.perf_counter() - t0) * 1e3
        for _ in range(12):        # the layers time themselves on their candidate buffers (three calls each) and keep the fastest
            step()
            torch.cuda.synchronize()
        # Calibration, still set-up: on a box whose free memory is a patchwork of physical blocks the arena's classes can be worse than
        # what the allocator hands out anyway (one of this round's six boxes: attention kernel 5.11 ms placed, 4.82 ms with fresh
        # allocations).  Ten placed steps against ten plain ones (x where torch put it, fresh outputs); the timed region runs the faster
        # configuration and the line says which (extras.placement.calibration).  extras.value_without_arena is measured either way.
        def _cal(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 10 * 1e3
        x_arena = x
        cal_placed = _cal(step)
        gcn.place_outputs = gat.place_outputs = False
        x = x_plain
        cal_plain = _cal(step)
        use_arena = ar0 is not None and cal_placed <= cal_plain
        if use_arena:
            gcn.place_outputs = gat.place_outputs = True
            x = x_arena
        placement_calibration = {"placed_ms_per_step": cal_placed, "plain_ms_per_step": cal_plain,
                                 "timed_region_runs": "placed (arena)" if use_arena else "plain allocations (the arena lost the calibration on this box)"}

    # ---- events around the two candidate dominant kernels (recorded on the launch stream) --------------------
    lib = L.load()
    step()
    out_p = torch.empty((N, D), dtype=torch.float32, device="cuda")
    Wx = gnnmp.dense(x, gat.dense_x_weight)
    a_hc = gat.a_hc               # keep alive: raw pointers below
    out_g = torch.empty((N, H * C), dtype=torch.float32, device="cuda")

    def k_propagate():
        L.check(lib.gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(x), None, L.ptr(c_slot), L.ptr(cvec),
                                              L.ptr(out_p), D, L.stream_ptr()))

    def k_gcn_fused():   # the GCN layer as the library runs it on this shape: aggregation + W + bias + relu in one kernel
        y = gnnmp.fused_conv(plan, L.SUM, x, gcn.weight, gcn.bias, "relu", ss_slot=c_slot, scale_dst=cvec)
        if y is None:     # small graphs (aggregate fits the Infinity Cache) are not fused by the library: time the layer as it runs
            gcn(g, x)

    def k_gat():
        L.check(lib.gnnmp_gat_conv_f32(plan.handle, L.ptr(Wx), None, L.ptr(a_hc), 0.2, L.ptr(gat.bias), L.ACT_RELU,
                                       L.ptr(out_g), H, C, L.stream_ptr()))

    for _ in range(args.warmup):
        step()
    # HIP events around the step's kernels INSIDE the timed region, on the stream they are launched on (gnnmp._lib.EventProbe: two
    # event records per launch, ~1 us each): roofline.avg_ms below is the in-step duration, the one a rocprofv3 kernel trace of this
    # command shows — not the kernel timed alone back to back (the two differed by 8 %)
    probe = L.EventProbe()
    dt, _ = timed_region(step, args.steps, 0, barrier, dist, before=lambda: L.set_probe(probe), after=lambda: L.set_probe(None))
    ms_per_step = dt / args.steps * 1e3
    value = world * 2 * Ep / (dt / args.steps)
    instep = {}
    for name in ("fused_conv", "gat_conv", "dense"):
        ts = sorted(probe.times_ms(name))
        if ts:
            instep[name] = {"ms": sum(ts) / len(ts), "ms_median": ts[len(ts) // 2], "launches": len(ts)}

    # per-kernel durations: HIP events on the stream the kernels are launched on (torch's current stream)
    def event_time(fn, iters):
        fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return sum(ts) / len(ts), ts[len(ts) // 2]

    iters = max(args.steps, 10)
    tp_avg, tp_med = event_time(k_propagate, iters)
    tg_avg, tg_med = event_time(k_gat, iters)
    tf_avg, tf_med = event_time(k_gcn_fused, iters)
    b_prop = alg_bytes_gcn_propagate(N, Ep, D)
    b_gat = alg_bytes_gat_aggregate(N, Ep, H, C)
    kern = {
        "gcn_propagate": {"kernel": "csr_rows_kernel", "ms": tp_avg, "ms_median": tp_med,
                          "alg_bytes": b_prop, "GBs": b_prop / tp_avg / 1e6,
                          "compulsory_bytes": compulsory_bytes(N, Ep, D, D)},
        # same algorithmic bytes as the propagate alone: the fused kernel reads what the propagate reads and writes the (N, Dout = D)
        # layer output instead of the (N, D) aggregate; the W image (40 KB per block) is not counted
        "gcn_fused_layer": {"kernel": "fused_conv_kernel" if args.workload == "products" else "csr_rows_kernel + dense_t16_kernel",
                            "ms": tf_avg, "ms_median": tf_med,
                            "alg_bytes": b_prop, "GBs": b_prop / tf_avg / 1e6,
                            "compulsory_bytes": compulsory_bytes(N, Ep, D, D),
                            "note": "includes the split rows' chunk pass + combine (two small launches before the fused kernel)"},
        "gat_aggregate": {"kernel": "gat_fused_rows_kernel", "ms": tg_avg, "ms_median": tg_med,
                          "alg_bytes": b_gat, "GBs": b_gat / tg_avg / 1e6,
                          "compulsory_bytes": compulsory_bytes(N, Ep, H * C, H * C) + 4 * N * H * C},
    }
    # the step's two kernels: the duration INSIDE the timed steps is the figure of record; the back-to-back one stays next to it
    for kname, pname in (("gcn_fused_layer", "fused_conv"), ("gat_aggregate", "gat_conv")):
        if pname in instep:
            k = kern[kname]
            k["alone_back_to_back_ms"] = k["ms"]
            k["ms"], k["ms_median"] = instep[pname]["ms"], instep[pname]["ms_median"]
            k["timed"] = f"HIP events around the launch inside the {args.steps} timed steps"
            k["GBs"] = k["alg_bytes"] / k["ms"] / 1e6
    for k in kern.values():   # the secondary line of SURVEY.md §8d: rate against the bytes an ideal cache could not avoid
        k["compulsory_GBs"] = k["compulsory_bytes"] / k["ms"] / 1e6
    kern["gcn_propagate"]["note"] = "the unfused propagate kernel, timed for reference: the step runs gcn_fused_layer instead"
    # dominant kernel of the step = the longer of the two (they are within a few per cent of each other and swap places from
    # box to box); the other one is reported next to it with the same fields
    def roof(name):
        k = kern[name]
        traffic, src = traffic_from_profiles(args.workload, k["kernel"].split(" ")[0])
        r = {"bound": "hbm", "kernel": k["kernel"], "call": name, "achieved": k["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": k["GBs"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
             "alg_bytes_per_launch": k["alg_bytes"], "avg_ms": k["ms"], "compulsory_bytes": k["compulsory_bytes"],
             "compulsory_frac": k["compulsory_GBs"] / HBM_PEAK_GBS,
             # against what HBM3E delivers (MI355X_MICROARCH.md: ~6.3 TB/s achievable of the 8 TB/s peak `frac` is priced on)
             "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": k["GBs"] / HBM_ACHIEVABLE_GBS}
        if traffic:   # L2->fabric bytes actually requested per second (Infinity-Cache hits included): what the memory system serves
            r["traffic_GBs"] = traffic / k["ms"] / 1e6
            r["traffic_frac"] = r["traffic_GBs"] / HBM_PEAK_GBS
        # WHAT THESE NUMBERS ARE.  `achieved` = ALGORITHMIC bytes / time priced against the 8 TB/s HBM
        # spec.  `traffic` = L2->fabric request bytes (TCC_EA0_RDREQ x 128 B + WRITE_SIZE): requests the 256 MiB Infinity Cache serves
        # are IN it — no counter of this rocprofv3 sees behind the Infinity Cache (rocprofv3 -L on gfx950: TCC_EA0_* only; there is no
        # MALL / DF / UMC counter) — so neither figure is "HBM bytes", and frac_of_achievable can exceed 1 (a streaming copy has no
        # Infinity-Cache hits, a gather of power-law sources has).  `hbm_bytes_est` is the model estimate of what DRAM serves.
        r["semantics"] = {"achieved": "algorithmic bytes per second (SURVEY.md 8d), not a measured HBM rate",
                          "traffic": "L2->fabric request bytes per launch, Infinity-Cache hits INCLUDED (not HBM bytes)",
                          "frac_of_achievable": "algorithmic GB/s over the 6.29 TB/s a streaming copy reaches; > 1 only because Infinity-Cache hits never reach HBM"}
        if traffic and name in mall:
            m = mall[name]
            rd = src_rw[name][0] if name in src_rw else traffic
            wr = traffic - rd
            est = rd * (1.0 - m["hit_share_upper_bound"]) + wr
            r["hbm_bytes_est"] = int(est)
            r["hbm_bytes_est_method"] = {
                "what": "read traffic x (1 - h) + write traffic, h = share of the row gathers that go to the hottest source rows that fit "
                        "the 256 MiB Infinity Cache (perfect-LFU bound from this graph's out-degree histogram): an UPPER bound on the hit "
                        "rate, hence a LOWER estimate of HBM bytes; the truth lies between hbm_bytes_est and traffic",
                "rows_resident": m["rows_resident"], "row_bytes": m["row_bytes"], "hit_share_upper_bound": m["hit_share_upper_bound"],
                "evidence": "profiles/r06_mall_sweep.txt: the bare row gather by working set (64 MB .. 2 GB)"}
            r["hbm_GBs_est"] = est / k["ms"] / 1e6
            r["hbm_frac_est"] = r["hbm_GBs_est"] / HBM_PEAK_GBS
        return r

    # Infinity-Cache residency model for the two gathered matrices: sources by decreasing out-degree (+1 for the self loop), as many rows
    # as fit 256 MiB, the share of all gathers they serve
    mall, src_rw = {}, {}
    try:
        outdeg = torch.bincount(sd - 1, minlength=N).to(torch.float64) + 1.0
        srt, _ = torch.sort(outdeg, descending=True)
        cum = torch.cumsum(srt, 0)
        for name, row_bytes in (("gcn_fused_layer", 4 * D), ("gat_aggregate", 4 * H * C)):
            rows = min(N, (256 << 20) // row_bytes)
            mall[name] = {"rows_resident": int(rows), "row_bytes": row_bytes, "hit_share_upper_bound": float(cum[rows - 1] / cum[-1])}
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            _doc = json.load(f).get(args.workload, {})
        for name, kn in (("gcn_fused_layer", "fused_conv_kernel"), ("gat_aggregate", "gat_fused_rows_kernel")):
            if kn in _doc:
                src_rw[name] = (_doc[kn]["hbm_read_bytes"], _doc[kn]["hbm_write_bytes"])
        del outdeg, srt, cum
    except Exception as e:   # the model is a side figure: never at the price of the line
        log(f"[bench] Infinity-Cache model skipped: {e!r}")

    step_kernels = [k for k in kern if k != "gcn_propagate"]
    dom = max(step_kernels, key=lambda k: kern[k]["ms"])
    roofline = roof(dom)
    roofline["other_step_kernel"] = roof([k for k in step_kernels if k != dom][0])

    # layer-level split and the other configs of BASELINE.json as side lines
    def layer_time(fn, iters, warm=1):
        # (sub-millisecond layers: 10 warm-up calls and 300 timed ones — the first launches after a synchronisation run before the clocks
        # and the launch pipeline have settled: 50 calls of a 0.15 ms layer read 6 % high against tools/small_configs.py's 300)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    extras = {"plan_create_ms": plan_ms, "norm_cache_ms": norm_cache_ms, "kernels": kern,
              "gcn_layer_ms": layer_time(lambda: gcn(g, x), 5), "gat_layer_ms": layer_time(lambda: gat(g, x), 5),
              "max_in_degree": plan.max_degree, "long_rows": plan.n_long}
    extras["in_step_kernels"] = instep
    # What is amortised ("weak"): the plan and GCN's normalisation cache are per-GRAPH work outside the timed region.  A caller
    # that meets a new graph — or passes `edge_weight` per call, which invalidates the norm cache — pays them inside the call: the first
    # call of each layer on a brand-new GNNGraph of the same edges, synchronised (plan build + caches + the layer itself).
    if rank == 0 and not args.no_extras:
        cold = {}
        for name, layer in (("gcn", gcn), ("gat", gat)):
            g_new = gnnmp.GNNGraph(sd, td, num_nodes=N, _validated=True)
            torch.cuda.synchronize(); t0c = time.perf_counter()
            layer(g_new, x)
            torch.cuda.synchronize()
            cold[name + "_first_call_on_a_new_graph_ms"] = (time.perf_counter() - t0c) * 1e3
            del g_new
        cold["what"] = ("plan build (dst-sorted CSR of the Int64 COO + self loops) + per-graph caches + the layer, one synchronised call on a new "
                        "GNNGraph: what `value` amortises over the calls on one graph")
        extras["cold_call"] = cold
    # L2->fabric bytes per launch of the OTHER configs' kernels from the committed counter passes (profiles/pmc_traffic.json: config 4's
    # mean aggregation, the arxiv pair, the chain kernel) next to the figures this run measures — constants of the profile, with provenance
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            _pt = json.load(f)
        extras["traffic_other_configs"] = {wl: {k: {"read_bytes": v["hbm_read_bytes"], "write_bytes": v["hbm_write_bytes"], "round": v.get("round")}
                                                for k, v in _pt.get(wl, {}).items()} for wl in ("sage", "arxiv", "batched")}
    except (OSError, ValueError):
        pass
    ar = gnnmp.placement.arena() if not args.no_placement else None
    extras["placement"] = {"enabled": ar is not None, "arena": (ar.info() if ar is not None else None),
                           "arena_create_and_first_step_ms": (arena_ms if not args.no_placement else None),
                           "class_of_x": (ar.class_of(x) if ar is not None else None),
                           "calibration": (placement_calibration if not args.no_placement else None),
                           "probe_budget": {"max_probe_gib": float(os.environ.get("GNNMP_ARENA_PROBE_GIB", "0")) or 32.0,
                                            "budget_ms": float(os.environ.get("GNNMP_ARENA_BUDGET_MS", "300"))},
                           "trials_ms": {name: {k[0]: ch.times_ms for k, ch in getattr(layer, "_placed", {}).items() if hasattr(ch, "times_ms")}
                                         for name, layer in (("gcn", gcn), ("gat", gat))},
                           "note": "outputs of the two gather kernels live in a placement class other than their gathered matrix's "
                                   "(csrc/arena.hip: 2 GiB blocks classified by a 0.6 ms probe inside a budget of 32 GiB held / ~0.3 s; fewer "
                                   "than two classes found = no placement, ordinary allocations)",
                           "no_arena_because": (gnnmp.placement._arena[2] if ar is None and len(gnnmp.placement._arena) > 2 else None)}
    if not args.no_placement:
        gcn.place_outputs = gat.place_outputs = False
        extras["placement"]["gcn_layer_ms_fresh_allocations"] = layer_time(lambda: gcn(g, x), 5)
        extras["placement"]["gat_layer_ms_fresh_allocations"] = layer_time(lambda: gat(g, x), 5)
        # the SAME timed region with nothing placed: features where torch's allocator put them, fresh output allocations per layer call —
        # what a caller gets who does not opt into gnnmp.placement / gnnmp_arena_* (the headline depends on the arena)
        def step_plain():
            return gcn(g, x_plain), gat(g, x_plain)
        for _ in range(args.warmup):
            step_plain()
        dt_plain, _ = timed_region(step_plain, args.steps, 0, barrier, dist)
        extras["value_without_arena"] = {"value": world * 2 * Ep / (dt_plain / args.steps), "unit": "edges/s",
                                         "ms_per_step": dt_plain / args.steps * 1e3, "steps": args.steps,
                                         "what": "the step of `value` with placement off: x in an ordinary allocation, fresh outputs per call"}
        gcn.place_outputs = gat.place_outputs = use_arena
    else:
        extras["value_without_arena"] = {"value": value, "unit": "edges/s", "ms_per_step": ms_per_step, "steps": args.steps,
                                         "what": "--no-placement: `value` itself is the un-placed figure"}
    extras["gcn_layer_edges_per_s"] = Ep / extras["gcn_layer_ms"] * 1e3
    extras["gat_layer_edges_per_s"] = Ep / extras["gat_layer_ms"] * 1e3
    if not args.no_extras:
        # SURVEY.md §8d's timing protocol next to the driver's K/W: >= 20 warm-up, >= 100 timed steps, per-step HIP events, median
        for _ in range(max(0, 20 - args.warmup)):
            step()
        _, med = event_time(step, 100)
        extras["protocol_8d"] = {"warmup": max(20, args.warmup), "timed_steps": 100, "median_ms_per_step": med,
                                 "edges_per_s_at_median": world * 2 * Ep / med * 1e3}
    # the dense contractions (the only MFMA work on the path): events around gnnmp_dense_f32, against the fp32 MFMA peak
    MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, 64 FLOP/clk/SIMD
    t_dg, _ = event_time(lambda: gnnmp.dense(out_p, gcn.weight, gcn.bias, "relu"), iters)
    t_da, _ = event_time(lambda: gnnmp.dense(x, gat.dense_x_weight), iters)
    # the bare ceiling of this box: a kernel that only fetches E' random rows into registers (tools/ubench/gather_probe.hip),
    # 26 rows per lane group (the mean in-degree) and 208; the step's two gather kernels are to be read against these
    if rank == 0 and not args.no_extras:
        try:
            import ctypes
            probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "ubench", "libgather_probe.so"))
            probe.gather_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            ids = torch.randint(0, N, (Ep,), device="cuda", dtype=torch.int32)
            ceil = {}
            for Dp in (D, H * C):
                xp = torch.randn((N, Dp), device="cuda")
                for per in (26, 208):
                    outp = torch.empty(((Ep + per - 1) // per, Dp), device="cuda")
                    ceil[f"D{Dp}_rows{per}_ms"] = event_time(
                        lambda: probe.gather_probe(xp.data_ptr(), ids.data_ptr(), Ep, 5, Dp, per, 8, outp.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream), 5)[0]
                del xp, outp
            del ids
            ceil["note"] = ("ms to fetch E' uniformly random rows of an N x D matrix into registers and nothing else; "
                            "compare kernels.gcn_fused_layer (D=%d) and kernels.gat_aggregate (D=%d)" % (D, H * C))
            extras["gather_ceiling"] = ceil
        except OSError as e:
            extras["gather_ceiling"] = {"note": f"tools/ubench/libgather_probe.so not built: {e}"}
    # the dense contractions: at this size they are bound by HBM requests, not by the matrix pipe (profiles/: the pipe is ~50 % busy) —
    # rated on the bytes they must move (x in, y out) against the HBM peak; the fp32-equivalent TF/s next to it, against the bf16 pipe's
    # 2.5 PF / 6 MFMAs per product (split-bf16) or the fp32 MFMA peak (dense_t16)
    def dense_line(shape, ms, K, Dout, kernel):
        b = 4.0 * N * (K + Dout)
        tf = 2.0 * N * K * Dout / ms / 1e9
        pipe_peak = 2500.0 / 6.0 if kernel.startswith("dense_split") else MFMA_F32_PEAK_TF
        return {"shape": shape, "kernel": kernel, "ms": ms, "hbm_bytes": b, "GBs": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS,
                "bound": "hbm", "TFs_fp32_equivalent": tf, "matrix_pipe_peak_TFs": pipe_peak, "matrix_pipe_frac": tf / pipe_peak}
    extras["dense"] = {
        "gcn_W_x": dense_line(f"{N}x{D}=>{D}", t_dg, D, D, "dense_split_kernel (split-bf16)" if ((D + 31) // 32 * 32) * 10 <= D * 11 else "dense_t16_kernel (fp32 MFMA 16x16x4)"),
        "gat_dense_x": dense_line(f"{N}x{D}=>{H * C}", t_da, D, H * C, "dense_split_kernel (3-plane split-bf16, six bf16 MFMAs per product, fp32 accumulate)"),
        "in_step": instep.get("dense")}
    if rank == 0 and not args.no_extras and args.workload == "products":
        try:      # side lines only: whatever goes wrong here must not take the bench line (or, at N > 1, the other ranks) with it
            # BASELINE.json config 4: SAGEConv(100 => 256) on the same graph (no self loops), aggr = mean and aggr = +
            sage = gnnmp.SAGEConv((D, 256), "relu", aggr="mean", seed=13)
            sage.place_outputs = sage.persistent_out = not args.no_placement   # (aggregate AND output placed: the second is its own opt-in)
            # (eight warm-up calls: the layer's placement trials — and, for the 2.5 GB output, the arena's classification of a buffer of its
            # own, tens of ms per class it has to refuse — are set-up and must be over before the five timed calls)
            t_sm = layer_time(lambda: sage(g, x), 5, warm=8)
            sage.aggr = "+"
            t_ss = layer_time(lambda: sage(g, x), 5, warm=2)
            extras["sage_products"] = {"E": E, "layer_ms_mean": t_sm, "layer_ms_sum": t_ss,
                                       "edges_per_s_mean": E / t_sm * 1e3, "edges_per_s_sum": E / t_ss * 1e3,
                                       "placed_buffers": sorted(k[0] for k, v in getattr(sage, "_placed", {}).items() if v is not None and v is not False)}
            del out_p, out_g, Wx, sage
            # SURVEY §8f "next" rows on the same graph: the standalone neighbourhood softmax (a20) and the training step of the two
            # headline layers (f1: forward + backward through the HIP adjoints)
            from gnnmp.backward import gat_conv_ad, gcn_conv_ad
            e8 = torch.randn((E, H), device="cuda")
            t_sx = layer_time(lambda: gnnmp.softmax_edge_neighbors(g, e8), 5)
            del e8
            xr = x.clone().requires_grad_(True)
            dyg = torch.randn((N, D), device="cuda")
            dya = torch.randn((N, H * C), device="cuda")
            for prm in (gcn.weight, gcn.bias, gat.dense_x_weight, gat.a, gat.bias):
                prm.requires_grad_(True)

            def train(fn, l, dy):
                def f():
                    fn(l, g, xr).backward(dy)
                    xr.grad = None
                return f

            def median_time(fn, iters=5):   # two warm-ups (the first calls build the reversed-edge plans and grow the allocator)
                fn(); fn()
                torch.cuda.synchronize()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
                for a_, b_ in ev:
                    a_.record(); fn(); b_.record()
                torch.cuda.synchronize()
                return sorted(a_.elapsed_time(b_) for a_, b_ in ev)[iters // 2]

            t_tg = median_time(train(gcn_conv_ad, gcn, dyg))
            t_ta = median_time(train(gat_conv_ad, gat, dya))
            for prm in (gcn.weight, gcn.bias, gat.dense_x_weight, gat.a, gat.bias):
                prm.requires_grad_(False)
                prm.grad = None
            extras["next_rows"] = {"softmax_edge_neighbors_h8_ms": t_sx, "gcn_fwd_bwd_ms": t_tg, "gat_fwd_bwd_ms": t_ta}
            del xr, dyg, dya
            # configs 2 and 3: arxiv shape
            Na, Da = synth.ARXIV["N"], synth.ARXIV["D"]
            sa, ta = synth.arxiv_like()
            ga = gnnmp.GNNGraph(torch.from_numpy(sa).cuda(), torch.from_numpy(ta).cuda(), num_nodes=Na, _validated=True)
            xa = torch.from_numpy(synth.features(Na, Da, seed=1)).cuda()
            gcn_a = gnnmp.GCNConv((Da, Da), "relu", seed=11)
            gat_a = gnnmp.GATConv((Da, C), "relu", heads=H, seed=12)
            Epa = len(sa) + Na
            tga = layer_time(lambda: gcn_a(ga, xa), 300, warm=10)
            taa = layer_time(lambda: gat_a(ga, xa), 300, warm=10)
            extras["arxiv"] = {"E_prime": Epa, "gcn_layer_ms": tga, "gat_layer_ms": taa,
                               "gcn_edges_per_s": Epa / tga * 1e3, "gat_edges_per_s": Epa / taa * 1e3}
            del ga, xa
            # config 5 on this one GPU: 8192 graphs, GraphConv x2 + GlobalPool(mean) + Dense — one fused launch (csrc/graph_chain2.hip)
            bstep, Gb, nb, eb = batched_setup(0, 1, None)
            tb = layer_time(bstep, 300, warm=10)
            extras["batched"] = {"graphs": Gb, "nodes": nb, "edges": eb, "ms_per_step": tb,
                                 "graphs_per_s": Gb / tb * 1e3, "edges_per_s": 2 * eb / tb * 1e3,
                                 "per_batch_prep_outside_the_timed_steps": getattr(batched_setup, "prep_ms", None),
                                 "new_batch_every_step": getattr(batched_setup, "new_batch", None),
                                 "training_step": getattr(batched_setup, "train", None)}
            del bstep
        except Exception as e:
            extras["side_lines_error"] = repr(e)

    # The one path of BASELINE.json that shards (config 5): when the driver runs the default workload on N > 1 ranks, the same N ranks
    # also run the graph-parallel step — shard by graph, ONE all-gather of the (G_r, 2) logits over RCCL, strong scaling — so that a
    # scaling run records the collective without a flag the driver does not pass.  At N = 1 the field carries world = 1.
    extras["rccl"] = {"world": world, "backend": (dist.get_backend() if dist is not None else None),
                      "collective": "all_gather_into_tensor of (gmax, 2) fp32 logits per rank, once per step" if world > 1 else None}
    if world > 1 and not args.no_extras:
        del x, g, plan
        torch.cuda.empty_cache()
        bstep, Gb, nb, eb = batched_setup(rank, world, dist)
        dtb, outb = timed_region(bstep, 100, 10, barrier, dist)
        dtb /= 100
        assert outb.shape == (Gb, 2)
        extras["batched_strong"] = {"graphs": Gb, "world": world, "ms_per_step": dtb * 1e3, "graphs_per_s": Gb / dtb,
                                    "scaling": "strong", "parallelism": f"graph-parallel x{world}: shard by graph, one all-gather of "
                                    f"(G_r, 2) logits per step (gnnmp.parallel.ShardPlan: no host work in the step)"}

    result = {
        "metric": ("edges/sec (fwd) GCNConv+GATConv, ogbn-products-shape; achieved HBM GB/s vs peak" if args.workload == "products" else
                   "edges/sec (fwd) GCNConv+GATConv, ogbn-arxiv-shape (BASELINE.json configs 2 + 3; not the headline metric)"),
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}-shape power-law graph N={N} E={E} (+N self loops, E'={Ep}), D={D}: "
                               f"GCNConv({D}=>{D},relu) fwd + GATConv({D}=>{C},heads={H},relu) fwd per step; "
                               f"edges counted = 2*E' per GPU",
                   "parallelism": f"replicas x{world} (independent feature batches, no collective)",
                   "index": "Int64 1-based COO as held by GNNGraph; plan = dst-sorted CSR (4-byte slots) built once",
                   "output_buffers": ("input features and layer outputs in the placement arena: every gather kernel's output in a placement class other "
                                      "than its gathered matrix's (gnnmp/placement.py, csrc/arena.hip); outputs persistent per layer"
                                      if (not args.no_placement and args.workload == "products" and use_arena) else
                                      "fresh allocation per call" + ("" if args.no_placement else " (the placement arena lost its calibration against plain allocations on this box: extras.placement.calibration)")),
                   "arithmetic": "fp32 operands, accumulation and results; aggregation in fp32 in the reference's edge order; the dense "
                                 "contraction of the GAT layer (dense_x) runs on the bf16 matrix core as an exact 3-plane split of every fp32 "
                                 "operand (six bf16 MFMAs per product, fp32 accumulate, dropped terms < 2^-21 of a product, truncation "
                                 "split: a one-signed bias of that size), GCN's W through fp32 MFMA"},
        "roofline": roofline,
        "extras": extras,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline((s, t, x_host) if keep_host else None, full_gat=args.cpu_baseline_full)
            ref = julia_reference_baseline()
            if ref is not None and "value" in ref:      # the reference itself ran here: it is the baseline, the port stays next to it
                ref["port"] = {k: result["cpu_baseline"][k] for k in ("value", "unit", "cores", "sample")}
                ref.setdefault("cpu_model", host_cpu_model())
                result["cpu_baseline"] = ref
            elif ref is not None:
                result["cpu_baseline"]["julia_reference"] = ref
        except Exception as e:  # the baseline must never take the bench line down
            result["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
