#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag> [quick]     e.g. r02  ->  gpurun_out/<tag>_*.csv
# One kernel-trace pass for durations, then one PMC pass per counter group (kernel-trace only, as the pool requires).
set -u
TAG=${1:-r02}
QUICK=${2:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extras"
run() {  # name, rocprof args..., -- command in $CMD
    local name=$1; shift
    rm -rf "/tmp/prof_$name"
    timeout 600 rocprofv3 "$@" -d "/tmp/prof_$name" -- $CMD > "$OUT/${TAG}_$name.log" 2>&1
    local db
    db=$(find "/tmp/prof_$name" -name '*_results.db' | head -1)
    if [ -n "$db" ]; then python "$REPO/tools/rocpd_summary.py" "$db" > "$OUT/${TAG}_$name.csv"; else echo "no db for $name" >&2; fi
}
CMD="$BENCH --steps 20 --warmup 5"
run bench_products_kernel_stats --kernel-trace --stats
CMD="python $REPO/tools/small_configs.py arxiv"
run arxiv_kernel_stats --kernel-trace --stats
CMD="python $REPO/tools/small_configs.py batched"
run batched_kernel_stats --kernel-trace --stats
CMD="python $REPO/tools/small_configs.py sage noplace"
run sage_kernel_stats --kernel-trace --stats
CMD="python $REPO/tools/experiments/batched_newbatch.py 8192 100"
run newbatch_kernel_stats --kernel-trace --stats
CMD="python $REPO/tools/backward_bench.py"
run backward_kernel_stats --kernel-trace --stats
CMD="python $REPO/tools/next_rows_bench.py --quick"
run next_rows_kernel_stats --kernel-trace --stats
if [ -z "$QUICK" ]; then
CMD="$BENCH --steps 3 --warmup 1"
run pmc_rd --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum
run pmc_wr --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
run pmc_fs --kernel-trace --pmc FETCH_SIZE
run pmc_ws --kernel-trace --pmc WRITE_SIZE
# the other configs' dominant kernels through the same three counter passes (round 5: csr_rows_kernel's mean aggregation of config 4,
# the arxiv pair of configs 2 / 3, the chain kernel of config 5) -> profiles/pmc_traffic.json via tools/pmc_traffic.py
for W in sage arxiv batched; do
  if [ "$W" = "sage" ]; then CMD="python $REPO/tools/small_configs.py sage noplace"; else CMD="python $REPO/tools/small_configs.py $W"; fi
  run pmc_${W}_rd --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum
  run pmc_${W}_fs --kernel-trace --pmc FETCH_SIZE
  run pmc_${W}_ws --kernel-trace --pmc WRITE_SIZE
done
CMD="python $REPO/tools/small_configs.py sage noplace"
run sage_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
CMD="python $REPO/tools/small_configs.py batched"
run batched_sq1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
CMD="python $REPO/tools/train_step_batched.py 8192 30"
run train_kernel_stats --kernel-trace --stats
CMD="python $REPO/tools/dense_small.py shape=2449029,100,100"
run dense_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
CMD="python $REPO/tools/dense_small.py shape=2449029,100,128"
run dense128_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
# round 6: fused_cat_kernel (knob 14 forces it) next to the two-kernel path of the same layer — durations, then the requests that reach
# the L2 (TCC_REQ: L2 hits included — the W image it streams) and those that leave it (TCC_EA0_RDREQ)
CMD="python $REPO/tools/small_configs.py sage noplace 14=16"
run sage_fused_kernel_stats --kernel-trace --stats
run pmc_sage_fused_l2 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
CMD="python $REPO/tools/small_configs.py sage noplace"
run pmc_sage_l2 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
fi
ls -la "$OUT" | grep "$TAG"
