set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, cmd..., 
    local name=$1; shift
    rm -rf "/tmp/prof_$name"
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d "/tmp/prof_$name" -- "$@" > "$OUT/r02x_$name.log" 2>&1
    local db
    db=$(find "/tmp/prof_$name" -name '*_results.db' | head -1)
    if [ -n "$db" ]; then python "$REPO/tools/rocpd_summary.py" "$db" | grep -A40 "PMC counters" | grep "dense_t16" > "$OUT/r02x_$name.csv"; fi
}
run d100 python $REPO/tools/dense_small.py shape=2449029,100,100
run d100_nomem python $REPO/tools/dense_small.py shape=2449029,100,100 13=3
run d128 python $REPO/tools/dense_small.py shape=2449029,100,128
run d128_nomem python $REPO/tools/dense_small.py shape=2449029,100,128 13=3
run pmc2 python $REPO/tools/dense_small.py shape=2449029,100,100
cat $OUT/r02x_*.csv
grep knobs $OUT/r02x_*.log
