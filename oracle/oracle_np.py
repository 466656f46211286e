"""Independent numpy/scipy twin of the C oracle — TEST INFRASTRUCTURE ONLY (same import rules as oracle.py).

Written separately from gnn_oracle.c (different language, different library calls) so that a mistake in one
restatement shows up as a disagreement.  `np.add.at` / `np.maximum.at` are unbuffered and apply duplicates in
index order, i.e. the same sequential-k order as NNlib's CPU scatter loop.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def gather(x, idx):
    return np.asarray(x, np.float32)[np.asarray(idx, np.int64) - 1]


def scatter(aggr, src, idx, n=None):
    src = np.asarray(src, np.float32)
    idx = np.asarray(idx, np.int64) - 1
    if n is None:
        n = int(idx.max()) + 1 if idx.size else 0
    shape = (n,) + src.shape[1:]
    if aggr in ("+", "sum", 0):
        out = np.zeros(shape, np.float32)
        np.add.at(out, idx, src)
    elif aggr in ("mean", 1):
        out = np.zeros(shape, np.float32)
        np.add.at(out, idx, src)
        cnt = np.zeros(n, np.int64)
        np.add.at(cnt, idx, 1)
        c = cnt.reshape((n,) + (1,) * (src.ndim - 1)).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            out = np.float32(0) + np.where(c == 0, out, out / c)
        out = out.astype(np.float32)
    elif aggr in ("max", 2):
        out = np.full(shape, -np.inf, np.float32)
        np.maximum.at(out, idx, src)
    elif aggr in ("min", 3):
        out = np.full(shape, np.inf, np.float32)
        np.minimum.at(out, idx, src)
    else:
        raise ValueError(aggr)
    return out


def propagate(aggr, s, t, n, xj, w=None, n_dst=None):
    m = gather(xj, s)
    if w is not None:
        m = np.asarray(w, np.float32).reshape((-1,) + (1,) * (m.ndim - 1)) * m
    return scatter(aggr, m, t, n if n_dst is None else n_dst)


def spmm_scipy(s, t, n, x, w=None):
    """x * A with A = sparse(s, t, w, n, n) via scipy (float64 accumulate): a tolerance-level cross-check only."""
    s = np.asarray(s, np.int64) - 1
    t = np.asarray(t, np.int64) - 1
    v = np.ones(len(s), np.float64) if w is None else np.asarray(w, np.float64)
    A = sp.coo_matrix((v, (s, t)), shape=(n, n)).tocsc()  # sums duplicates
    return (A.T @ np.asarray(x, np.float64)).astype(np.float64)


def softmax_edge_neighbors(t, n, e):
    e = np.asarray(e, np.float32)
    mx = gather(scatter("max", e, t, n), t)
    num = np.exp(e - mx).astype(np.float32)
    den = gather(scatter("+", num, t, n), t)
    return (num / den).astype(np.float32)


def degree(idx, n, w=None):
    src = np.ones(len(idx), np.float32) if w is None else np.asarray(w, np.float32)
    return np.zeros(n, np.float32) + scatter("+", src, idx, n)
