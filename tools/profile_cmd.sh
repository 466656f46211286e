#!/bin/bash
# rocprofv3 kernel statistics of one command:  tools/profile_cmd.sh <name> <command...>  ->  gpurun_out/<name>_kernel_stats.csv
set -u
NAME=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$NAME"
( cd "$REPO" && timeout 900 rocprofv3 --kernel-trace --stats -d "/tmp/prof_$NAME" -- "$@" > "$OUT/${NAME}.log" 2>&1 )
db=$(find "/tmp/prof_$NAME" -name '*_results.db' | head -1)
if [ -n "$db" ]; then python "$REPO/tools/rocpd_summary.py" "$db" > "$OUT/${NAME}_kernel_stats.csv"; else echo "no db for $NAME" >&2; fi
head -12 "$OUT/${NAME}_kernel_stats.csv"
