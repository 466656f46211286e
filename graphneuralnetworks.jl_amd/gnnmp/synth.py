"""Seeded synthetic workloads of BASELINE.json's configs (no datasets are reachable: `data: synthetic`).

numpy only (so the same arrays feed the CPU oracle and, after upload, the HIP path).  Indices are returned 1-based
int64, exactly as a Julia GNNGraph holds them.  Sizes follow SURVEY.md §8:
  arxiv     N=169 343, E=1 166 243 directed: uniform sources, power-law destinations (max in-degree ~1e4), no self loops
  products  N=2 449 029, E=61 859 140 = 2 x 30 929 570 bidirected pairs, power-law both ends (max degree ~1.7e4)
  batched   G graphs, n_i ~ U{20..40}, m_i = 4 n_i directed edges (bidirected pairs, no self loops), rand_graph-like
  cora      N=2 708, E=10 556 (bidirected), 1433 sparse row-normalised features
"""
from __future__ import annotations

import numpy as np

ARXIV = dict(N=169_343, E=1_166_243, D=128)
PRODUCTS = dict(N=2_449_029, E=61_859_140, D=100)
CORA = dict(N=2_708, E=10_556, D=1_433, C=7)


def _powerlaw_nodes(rng, N, M, alpha, perm):
    """M node ids with P(rank r) ~ r^(1/alpha - 1); ranks are decorrelated from ids by `perm`."""
    u = rng.random(M)
    r = np.minimum((N * u ** alpha).astype(np.int64), N - 1)
    return perm[r]


def arxiv_like(N=ARXIV["N"], E=ARXIV["E"], alpha=2.67, seed=0):
    """directed citation-like graph: (s, t) 1-based int64"""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(N)
    s = rng.integers(0, N, size=E, dtype=np.int64)
    t = _powerlaw_nodes(rng, N, E, alpha, perm)
    loop = s == t
    t[loop] = (t[loop] + 1) % N
    return s + 1, t + 1


def products_like(N=PRODUCTS["N"], E=PRODUCTS["E"], alpha=1.79, seed=2):
    """bidirected co-purchase-like graph: E/2 pairs, both directions, 1-based int64"""
    assert E % 2 == 0
    rng = np.random.default_rng(seed)
    perm = rng.permutation(N)
    M = E // 2
    u = _powerlaw_nodes(rng, N, M, alpha, perm)
    v = _powerlaw_nodes(rng, N, M, alpha, perm)
    loop = u == v
    v[loop] = (v[loop] + 1) % N
    s = np.concatenate([u, v]) + 1
    t = np.concatenate([v, u]) + 1
    return s, t


def features(N, D, seed=1):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((N, D), dtype=np.float32)


def rand_graph(n, m, rng):
    """rand_graph(n, m; bidirected=true) semantics (GNNGraphs/src/generate.jl:41-65): m directed edges = m/2 distinct
    undirected pairs without self loops, listed as (u,v) then (v,u).  1-based int64."""
    assert m % 2 == 0
    pairs = set()
    while len(pairs) < m // 2:
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        if a == b:
            continue
        pairs.add((min(a, b), max(a, b)))
    p = np.array(sorted(pairs), np.int64).reshape(-1, 2)
    s = np.concatenate([p[:, 0], p[:, 1]]) + 1
    t = np.concatenate([p[:, 1], p[:, 0]]) + 1
    return s, t


def batched_graphs(G=8192, nmin=20, nmax=40, deg=4, seed=3):
    """list of (s, t, n) member graphs (local 1-based numbering), sizes n_i ~ U{nmin..nmax}, m_i = deg * n_i.
    Vectorised: pairs are drawn per graph from a ring + random chords so that no duplicate / self loop appears."""
    rng = np.random.default_rng(seed)
    ns = rng.integers(nmin, nmax + 1, size=G)
    out = []
    for n in ns:
        n = int(n)
        # deg/2 undirected edges per node: ring offsets 1..deg/2 give exactly n*deg/2 distinct pairs, no loops;
        # a random relabelling makes it a random-looking regular graph.
        lab = rng.permutation(n)
        u = np.concatenate([np.arange(n)] * (deg // 2))
        off = np.repeat(np.arange(1, deg // 2 + 1), n)
        v = (u + off) % n
        a, b = lab[u], lab[v]
        s = np.concatenate([a, b]).astype(np.int64) + 1
        t = np.concatenate([b, a]).astype(np.int64) + 1
        out.append((s, t, n))
    return out


def cora_like(seed=17):
    """Cora-shaped graph + features (GNNGraphs/src/mldatasets.jl:14-21 shapes): bidirected, no self loops."""
    N, E, D = CORA["N"], CORA["E"], CORA["D"]
    rng = np.random.default_rng(seed)
    perm = rng.permutation(N)
    M = E // 2
    u = _powerlaw_nodes(rng, N, M, 1.6, perm)
    v = rng.integers(0, N, size=M, dtype=np.int64)
    loop = u == v
    v[loop] = (v[loop] + 1) % N
    s = np.concatenate([u, v]) + 1
    t = np.concatenate([v, u]) + 1
    x = (rng.random((N, D)) < 0.0127).astype(np.float32)
    rs = x.sum(axis=1, keepdims=True)
    x = x / np.maximum(rs, 1.0)
    return s, t, x.astype(np.float32)
