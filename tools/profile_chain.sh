#!/bin/bash
# rocprofv3 evidence for the batched config (BASELINE.json config 5): kernel durations + SQ counters of the chain kernels
#   tools/profile_chain.sh <tag>   ->  gpurun_out/<tag>_batched_*.csv
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    rm -rf "/tmp/prof_$name"
    timeout 600 rocprofv3 "$@" -d "/tmp/prof_$name" -- $CMD > "$OUT/${TAG}_$name.log" 2>&1
    local db
    db=$(find "/tmp/prof_$name" -name '*_results.db' | head -1)
    if [ -n "$db" ]; then python "$REPO/tools/rocpd_summary.py" "$db" > "$OUT/${TAG}_$name.csv"; else echo "no db for $name" >&2; fi
}
CMD="python $REPO/tools/experiments/chain_bench.py"
run batched_kernel_stats --kernel-trace --stats
run batched_sq1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run batched_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
ls -la "$OUT" | grep "$TAG"
