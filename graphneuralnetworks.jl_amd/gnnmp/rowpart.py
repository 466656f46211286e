"""1-D destination-row partition of ONE large graph across GPUs (the "next" item for configs 2-4).

Each rank owns a contiguous range of destination rows chosen so that every rank holds about E/world edges, builds a
plan for ITS rows over the FULL source set, and computes its slice of every layer's output.  The exchange step is one
all-gather of the (N, D) layer output per layer (RCCL over xGMI on the GPU box, gloo in the CPU tests): 4*N*D/world
bytes leave every rank — none for a replicated layer-0 input.  §8e prices it: products 122 MB per rank ≈ 0.8 ms with a
direct all-peer all-gather against a 0.6 ms per-rank kernel, so this mode is communication-bound by design and is NOT
what `bench.py` reports by default (replicas); it exists for graphs whose features do not fit one GPU's work budget and
is covered by world-size-2/3 tests that require bit-identical results to the unpartitioned forward (rows are independent
units: a row's edges, and their order, are the same on whichever rank owns it).

Graph prep (the partition, the local edge lists) is integer work done once per graph with torch indexing; the data path
per layer is: local HIP forward on the rank's rows -> all_gather -> next layer.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch


def partition_rows_by_edges(t: torch.Tensor, num_nodes: int, world: int, index_base: int = 1) -> List[Tuple[int, int]]:
    """Contiguous row ranges [lo, hi) (0-based) with ~equal edge counts: boundary r is the first row whose cumulative
    in-degree reaches r*E/world.  Deterministic; every row belongs to exactly one range; ranges may be empty."""
    deg = torch.bincount((t.to(torch.int64) - index_base).cpu(), minlength=num_nodes)
    cum = torch.cumsum(deg, 0)
    E = int(cum[-1]) if num_nodes > 0 else 0
    cuts = [0]
    for r in range(1, world):
        target = (E * r + world - 1) // world
        k = int(torch.searchsorted(cum, torch.tensor(target), right=False)) + 1 if E > 0 else (num_nodes * r) // world
        cuts.append(min(max(k, cuts[-1]), num_nodes))
    cuts.append(num_nodes)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def local_edges(s: torch.Tensor, t: torch.Tensor, lo: int, hi: int, index_base: int = 1):
    """the edges whose destination lies in [lo, hi), in ORIGINAL relative order (so per-row summation order is unchanged);
    destinations renumbered to the local range, sources left global.  Also returns the kept edge positions (0-based)."""
    t0 = t.to(torch.int64) - index_base
    keep = torch.nonzero((t0 >= lo) & (t0 < hi)).flatten()
    return s[keep].contiguous(), (t[keep] - lo).contiguous(), keep


def gather_rows(local_out: torch.Tensor, bounds: Sequence[Tuple[int, int]], rank: int, world: int, dist=None) -> torch.Tensor:
    """the exchange step: every rank contributes its (hi - lo, D) slice, every rank ends with the full (N, D) matrix"""
    N = bounds[-1][1]
    D = local_out.shape[1]
    lo, hi = bounds[rank]
    assert local_out.shape[0] == hi - lo
    if world == 1 or dist is None:
        return local_out
    rmax = max(h - l for l, h in bounds)
    pad = torch.zeros((rmax, D), dtype=local_out.dtype, device=local_out.device)
    pad[: hi - lo] = local_out
    buf = torch.empty((world * rmax, D), dtype=local_out.dtype, device=local_out.device)
    if hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(buf, pad)          # ONE collective per layer
    else:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        buf = torch.cat(parts, 0)
    full = torch.empty((N, D), dtype=local_out.dtype, device=local_out.device)
    for r, (l, h) in enumerate(bounds):
        full[l:h] = buf[r * rmax: r * rmax + (h - l)]
    return full


def row_parallel_forward(layers: Sequence[Callable], x_full: torch.Tensor, bounds, rank: int, world: int, dist=None):
    """layers[k](x_full) -> this rank's (hi - lo, D_k) slice of layer k's output (a closure over the rank's local graph);
    between layers the slices are all-gathered into the next layer's replicated input.  Returns the full final output."""
    h = x_full
    for f in layers:
        h = gather_rows(f(h), bounds, rank, world, dist)
    return h
