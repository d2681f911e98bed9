#!/bin/bash
# Run on the GPU box (via gpurun): every rocprofv3 pass of the round, summaries under gpurun_out/<tag>/ ready for profiles/<tag>/.
#   scripts/profile_round.sh [tag]
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
bash scripts/profile_bench.sh "$TAG" > "$OUT/profile_bench.log" 2>&1
for f in bench_kernel_stats.csv bench_pmc_summary.txt bench_pmc.json bench_under_rocprof.json; do cp "gpurun_out/prof_$TAG/$f" "$OUT/" 2>/dev/null; done
bash scripts/profile_train.sh "$TAG" > "$OUT/profile_train.log" 2>&1
cp gpurun_out/prof_train_$TAG/*.csv gpurun_out/prof_train_$TAG/*.txt "$OUT/" 2>/dev/null
bash scripts/pmc_spline.sh > "$OUT/spline_pmc_summary.txt" 2>&1
ls -la "$OUT"
