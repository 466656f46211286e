"""Generates tests/golden/hotpath_v1.npz — small input/expected-output vectors for the hot path.

The reference is Julia and cannot run in the build container, so the expected outputs come from the CPU oracle
(oracle/gnn_oracle.c + oracle/oracle.py), which tests/test_oracle_reference_pins.py pins against the reference's own
known-answer tests.  The fixture is DATA (inputs + expected outputs); nothing of the reference's source is stored.

    python tests/golden/make_golden.py        # rewrites the fixture (deterministic: seeded)

Layout of the npz: keys "<case>/<array>".  Index arrays are 1-based int64; features float32 [N, D].
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def coo_from_adj(A):
    s, t = [], []
    n = A.shape[0]
    for j in range(n):
        for i in range(n):
            for _ in range(int(A[i, j])):
                s.append(i + 1)
                t.append(j + 1)
    return np.array(s, np.int64), np.array(t, np.int64)


def build_cases():
    cases = {}
    rng = np.random.default_rng(20241218)

    # ---- propagate on assorted small topologies, every aggr, weighted and not ---------------------------------
    topo = {}
    adj1 = np.array([[0, 1, 0, 1], [1, 0, 1, 0], [0, 1, 0, 1], [1, 0, 1, 0]])
    adj2 = np.array([[0, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 1], [1, 0, 1, 0]])
    topo["cycle4"] = coo_from_adj(adj1) + (4,)
    topo["isolated4"] = coo_from_adj(adj2) + (4,)
    n = 128
    A = (rng.random((n, n)) < 0.1).astype(np.int64)
    topo["n128"] = coo_from_adj(A) + (n,)
    # multi-edges + self loops + isolated nodes + a hub with a long row (> 512 edges: the split-row path)
    n = 700
    s = np.concatenate([rng.integers(1, n + 1, 900), np.arange(1, 651), [5, 5, 5, 7, 7]])
    t = np.concatenate([rng.integers(1, n - 20, 900), np.full(650, 3), [5, 5, 5, 7, 7]])
    p = rng.permutation(len(s))
    topo["hub700"] = (s[p].astype(np.int64), t[p].astype(np.int64), n)
    topo["empty5"] = (np.zeros(0, np.int64), np.zeros(0, np.int64), 5)

    for name, (s, t, n) in topo.items():
        for D in ((3, 16) if name != "n128" else (10, 32)):
            x = rng.standard_normal((n, D)).astype(np.float32)
            w = rng.random(len(s)).astype(np.float32)
            c = {"s": s, "t": t, "n": np.int64(n), "x": x, "w": w}
            for aggr in ("+", "mean", "max", "min"):
                key = {"+": "sum"}.get(aggr, aggr)
                c[f"out_{key}"] = orc.propagate(aggr, s, t, n, x)
                c[f"outw_{key}"] = orc.propagate(aggr, s, t, n, x, w)
            c["deg_in"] = orc.degree(t, n)
            c["deg_in_w"] = orc.degree(t, n, w)
            s2, t2, w2 = orc.add_self_loops(s, t, n, w)
            c["s_loops"], c["t_loops"], c["w_loops"] = s2, t2, w2
            H = 2
            e = (2 * rng.standard_normal((len(s), H))).astype(np.float32)
            c["logits"] = e
            c["alpha"] = orc.softmax_edge_neighbors(t, n, e)
            cases[f"prop_{name}_D{D}"] = c

    # ---- layers on TEST_GRAPHS-like graphs with fixed weights -------------------------------------------------
    D_IN, D_OUT = 3, 5
    for name in ("cycle4", "isolated4", "n128"):
        s, t, n = topo[name]
        x = rng.random((n, D_IN), dtype=np.float32)
        c = {"s": s, "t": t, "n": np.int64(n), "x": x}
        W = rng.standard_normal((D_OUT, D_IN)).astype(np.float32) * 0.5
        W2 = rng.standard_normal((D_OUT, D_IN)).astype(np.float32) * 0.5
        Ws = rng.standard_normal((D_OUT, 2 * D_IN)).astype(np.float32) * 0.5
        b = rng.standard_normal(D_OUT).astype(np.float32) * 0.1
        ew = rng.random(len(s)).astype(np.float32)
        c.update(W=W, W2=W2, Ws=Ws, b=b, ew=ew)
        c["gcn"] = orc.gcn_conv(s, t, n, x, W, b, "relu", blas=False)
        c["gcn_noloops"] = orc.gcn_conv(s, t, n, x, W, b, None, add_self_loops_=False, blas=False)
        c["gcn_ew"] = orc.gcn_conv(s, t, n, x, W, b, None, edge_weight=ew, blas=False)
        c["gcn_gw"] = orc.gcn_conv(s, t, n, x, W, b, None, use_edge_weight=True, graph_w=ew, blas=False)
        Wwide = rng.standard_normal((2, D_IN)).astype(np.float32)   # Dout < Din: W applied first (conv.jl:36-40)
        c["Wwide"] = Wwide
        c["gcn_wfirst"] = orc.gcn_conv(s, t, n, x, Wwide, b[:2], "relu", blas=False)
        for aggr in ("+", "mean", "max"):
            key = {"+": "sum"}.get(aggr, aggr)
            c[f"graphconv_{key}"] = orc.graph_conv(s, t, n, x, W, W2, b, "relu", aggr, blas=False)
            c[f"sage_{key}"] = orc.sage_conv(s, t, n, x, Ws, b, None, aggr, blas=False)
        for heads in (1, 2):
            Wd = rng.standard_normal((D_OUT * heads, D_IN)).astype(np.float32) * 0.5
            a = rng.standard_normal((2 * D_OUT, heads)).astype(np.float32) * 0.5
            c[f"gat_Wd_h{heads}"], c[f"gat_a_h{heads}"] = Wd, a
            for concat in (True, False):
                bb = rng.standard_normal(D_OUT * heads if concat else D_OUT).astype(np.float32) * 0.1
                y, alpha = orc.gat_conv(s, t, n, x, Wd, a, bb, "relu", heads, concat, blas=False, return_alpha=True)
                tag = f"h{heads}_{'cat' if concat else 'mean'}"
                c[f"gat_b_{tag}"], c[f"gat_{tag}"], c[f"gat_alpha_{tag}"] = bb, y, alpha
        cases[f"layers_{name}"] = c

    # ---- the reference's closed-form GCN case (GraphNeuralNetworks/test/layers/conv.jl:30-44) --------------------
    s = np.array([2, 3, 1, 3, 1, 2], np.int64)
    t = np.array([1, 1, 2, 2, 3, 3], np.int64)
    w = np.array([1, 2, 3, 4, 5, 6], np.float32)
    x = np.ones((3, 1), np.float32)
    y = orc.gcn_conv(s, t, 3, x, np.ones((1, 1), np.float32), np.zeros(1, np.float32),
                     add_self_loops_=False, use_edge_weight=True, graph_w=w, blas=False)
    cases["gcn_closed_form"] = {"s": s, "t": t, "w": w, "x": x, "y": y,
                                "y_expected_ref": np.array([0.5663732, 1.110496], np.float32)}

    # ---- batch of 5 graphs + GlobalPool ------------------------------------------------------------------------
    gs = []
    for i in range(5):
        n = int(rng.integers(3, 12))
        m = int(rng.integers(2, 3 * n))
        gs.append((rng.integers(1, n + 1, m).astype(np.int64), rng.integers(1, n + 1, m).astype(np.int64), n))
    s, t, gi, N = orc.batch(gs)
    x = rng.standard_normal((N, 16)).astype(np.float32)
    c = {"s": s, "t": t, "gi": gi, "n": np.int64(N), "x": x,
         "edge_ptr": np.array([0] + [len(g[0]) for g in gs], np.int64).cumsum(),
         "node_ptr": np.array([0] + [g[2] for g in gs], np.int64).cumsum(),
         "s_local": np.concatenate([g[0] for g in gs]), "t_local": np.concatenate([g[1] for g in gs])}
    for aggr in ("+", "mean", "max", "min"):
        c[f"pool_{ {'+': 'sum'}.get(aggr, aggr) }"] = orc.global_pool(aggr, gi, x, 5)
    W1 = rng.standard_normal((8, 16)).astype(np.float32) * 0.3
    W2 = rng.standard_normal((8, 16)).astype(np.float32) * 0.3
    b = rng.standard_normal(8).astype(np.float32) * 0.1
    h = orc.graph_conv(s, t, N, x, W1, W2, b, "relu", "+", blas=False)
    c.update(W1=W1, W2=W2, b=b, h=h, logits=orc.global_pool("mean", gi, h, 5))
    cases["batch5"] = c
    return cases


def main():
    cases = build_cases()
    flat = {}
    for cname, c in cases.items():
        for k, v in c.items():
            if v is None:
                continue
            flat[f"{cname}/{k}"] = np.asarray(v)
    out = os.path.join(HERE, "hotpath_v1.npz")
    np.savez_compressed(out, **flat)
    print(out, os.path.getsize(out), "bytes,", len(cases), "cases,", len(flat), "arrays")


if __name__ == "__main__":
    main()
