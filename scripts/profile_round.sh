#!/bin/bash
# Run on the GPU box (via gpurun): every rocprofv3 pass of the round, summaries under gpurun_out/<tag>/ ready for profiles/<tag>/.
#   scripts/profile_round.sh [tag]
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
bash scripts/profile_bench.sh "$TAG" > "$OUT/profile_bench.log" 2>&1
for f in bench_kernel_stats.csv bench_pmc_summary.txt bench_pmc.json bench_under_rocprof.json; do cp "gpurun_out/prof_$TAG/$f" "$OUT/" 2>/dev/null; done
bash scripts/profile_train.sh "$TAG" > "$OUT/profile_train.log" 2>&1
cp gpurun_out/prof_train_$TAG/*.csv gpurun_out/prof_train_$TAG/*.txt "$OUT/" 2>/dev/null
bash scripts/pmc_spline.sh > "$OUT/spline_pmc_summary.txt" 2>&1
ls -la "$OUT"
# the spline kernels' durations (bench.py's spline_flow.roofline / train_roofline cite these files)
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_spline_$TAG/mh" -- python3 $R/tools/time_spline.py > "$R/$OUT/spline_times_under_rocprof.txt" 2> /dev/null
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_spline_$TAG/train" -- python3 $R/tools/time_spline_train.py > "$R/$OUT/spline_train_times_under_rocprof.txt" 2> /dev/null )
f=$(find gpurun_out/prof_spline_$TAG/mh -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > "$OUT/spline_kernel_stats.csv"
f=$(find gpurun_out/prof_spline_$TAG/train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -10 "$f" > "$OUT/spline_train_kernel_stats.csv"
find gpurun_out/prof_spline_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_spline_$TAG -name "*agent_info.csv" -delete
ls -la "$OUT"
