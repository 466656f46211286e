"""Helper of tests/test_parallel_two_ranks_gpu.py: one rank of the graph-parallel config-5 step (bench.batched_setup: the HIP chain
kernel on this rank's shard, ShardPlan's device buffers, ONE all-gather, one index_select) with BOTH ranks on cuda:0 and gloo as the
transport — the GPU boxes of the pool have one GPU, RCCL needs one device per rank.  Rank 0 also runs the unsharded step and compares."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

import bench

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
step, g_total, nodes, edges = bench.batched_setup(rank, world, dist, G=G)
y = step()
for _ in range(3):
    assert torch.equal(step(), y), "the sharded step is not run-to-run identical"
assert y.shape == (G, 2) and y.is_cuda
gathered = [torch.empty_like(y) for _ in range(world)]
dist.all_gather(gathered, y)
assert all(torch.equal(v, y) for v in gathered), "ranks disagree on the gathered logits"
if rank == 0:
    ref_step, *_ = bench.batched_setup(0, 1, None, G=G)
    ref = ref_step()
    # member graphs are independent units and a row's arithmetic does not depend on its tile: regrouping changes no bit
    assert torch.equal(ref, y), float((ref - y).abs().max())
    print(f"TWO_RANKS_OK G={G} nodes={nodes} edges={edges}", flush=True)
dist.barrier()
dist.destroy_process_group()
