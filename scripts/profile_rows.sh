#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of repeated spline training calls (rows form by default; NNEST_SPL_ROWS=0 for the tile form).
#   scripts/profile_rows.sh [tag] [D]
set -u
TAG=${1:-r06}
DD=${2:-50}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_rows_${TAG}_d$DD" -- python3 $R/tools/time_spline_train.py $DD > "$R/$OUT/rows_train_times_under_rocprof_d$DD.txt" 2> /dev/null )
f=$(find gpurun_out/prof_rows_${TAG}_d$DD -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > "$OUT/rows_train_kernel_stats_d$DD.csv"
find gpurun_out/prof_rows_${TAG}_d$DD -name "*kernel_trace.csv" -delete; find gpurun_out/prof_rows_${TAG}_d$DD -name "*agent_info.csv" -delete
cat "$OUT/rows_train_times_under_rocprof_d$DD.txt"; cut -c1-160 "$OUT/rows_train_kernel_stats_d$DD.csv"
