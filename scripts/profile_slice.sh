#!/bin/bash
# rocprofv3 kernel-trace stats of the slice proposal kernel (tools/time_slice.py): the file behind `slice_proposal.roofline.profile`
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_slice_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
python3 $R/tools/time_slice.py > "$OUT/time_slice.txt" 2>&1
cat "$OUT/time_slice.txt"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/tools/time_slice.py > "$OUT/under_rocprof.txt" 2> "$OUT/rocprof.log"
for f in $(find "$OUT" -name "*kernel_stats.csv"); do head -6 "$f" | cut -c1-220 > "$OUT/slice_kernel_stats.csv"; cat "$OUT/slice_kernel_stats.csv"; done
