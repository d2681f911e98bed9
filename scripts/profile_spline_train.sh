#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + SQ counters of the spline training kernels
# (tools/time_spline_train.py: 1000 live points, x_dim 50, 8 calls of 40 epochs).  Summaries to gpurun_out/prof_spline_train_<tag>/.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_spline_train_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
python3 $R/tools/time_spline_train.py 50 > "$OUT/spline_train_epoch_times.txt" 2>&1
python3 $R/tools/time_spline_train.py 20 >> "$OUT/spline_train_epoch_times.txt" 2>&1
cat "$OUT/spline_train_epoch_times.txt"
bash $R/tools/trace_spline_train.sh $TAG 50 > "$OUT/spline_train_kernel_stats.txt" 2>&1
cat "$OUT/spline_train_kernel_stats.txt"
bash $R/tools/pmc_spline_train.sh 50 > "$OUT/spline_train_pmc_summary.txt" 2>&1
cat "$OUT/spline_train_pmc_summary.txt"
