"""Graph-parallel forward for batched graphs (BASELINE.json config 5).

A batched GNNGraph is block-diagonal (MLUtils.batch, GNNGraphs/src/transform.jl:682-709): no edge crosses member graphs,
so member graphs are independent units.  One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests):
  1. `shard_by_size` deals the member graphs to ranks, largest first, round-robin — balances nodes and edges;
  2. every rank batches ITS graphs locally (local node offsets, local graph_indicator) and runs the whole forward,
     including GlobalPool, producing (G_r, nout) logits;
  3. ONE all-gather of the per-shard logits (padded to the largest shard; a few KB per rank: latency-bound, so a single
     collective and no bucketing), then a permutation back to the caller's graph order.
Weights are replicated and read-only; there is no all-reduce, no halo exchange, nothing else on the data path.
The reference has no distributed code at all: this is the build's own design for north_star's
"shard by graph ... all-gather of per-shard logits".
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch


def shard_by_size(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Deterministic balanced assignment: graphs sorted by size (descending, index as tie-break) are dealt in a
    boustrophedon (snake) order so that every rank gets the same count +-1 and nearly the same total size."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    shards: List[List[int]] = [[] for _ in range(world)]
    for pos, gi in enumerate(order):
        rnd, k = divmod(pos, world)
        r = k if rnd % 2 == 0 else world - 1 - k
        shards[r].append(gi)
    for s in shards:
        s.sort()
    return shards


def gather_shard_outputs(local_out: torch.Tensor, shards: List[List[int]], rank: int, world: int, dist=None) -> torch.Tensor:
    """all-gather the (G_r, nout) outputs of every rank and put the rows back in the original graph order."""
    n_total = sum(len(s) for s in shards)
    nout = local_out.shape[1]
    assert local_out.shape[0] == len(shards[rank])
    if world == 1 or dist is None:
        gathered = [local_out]
    else:
        gmax = max(len(s) for s in shards)
        pad = torch.zeros((gmax, nout), dtype=local_out.dtype, device=local_out.device)
        pad[: local_out.shape[0]] = local_out
        buf = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(buf, pad)                      # the only collective on the path
        gathered = [buf[r][: len(shards[r])] for r in range(world)]
    out = torch.empty((n_total, nout), dtype=local_out.dtype, device=local_out.device)
    for r in range(len(gathered)):
        if len(shards[r]):
            idx = torch.as_tensor(shards[r], dtype=torch.long, device=local_out.device)
            out[idx] = gathered[r]
    return out


def graph_parallel_forward(member_graphs: Sequence, forward_local: Callable, rank: int, world: int, dist=None,
                           sizes: Sequence[int] = None):
    """member_graphs: any sequence of per-graph records; forward_local(list_of_records) -> (len(list), nout) tensor
    (it batches the records and runs the model on this rank's device).  Returns the (G, nout) outputs in the original
    order on every rank, and this rank's shard (for reporting)."""
    if sizes is None:
        sizes = [int(getattr(g, "num_nodes", 1)) for g in member_graphs]
    shards = shard_by_size(sizes, world)
    mine = [member_graphs[i] for i in shards[rank]]
    local_out = forward_local(mine)
    return gather_shard_outputs(local_out, shards, rank, world, dist), shards[rank]
