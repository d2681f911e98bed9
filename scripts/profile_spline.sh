#!/bin/bash
# rocprofv3 kernel-trace stats of the spline-flow kernels (tools/time_spline.py)
set -u
TAG=${1:-r01c}
OUT=$PWD/gpurun_out/prof_spline_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
python3 $R/tools/time_spline.py > "$OUT/time_spline.txt" 2>&1
cat "$OUT/time_spline.txt"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/tools/time_spline.py > "$OUT/under_rocprof.txt" 2> "$OUT/rocprof.log"
for f in $(find "$OUT" -name "*kernel_stats.csv"); do echo "== $f"; head -12 "$f" | cut -c1-160; done
