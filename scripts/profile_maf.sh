#!/bin/bash
# rocprofv3 kernel-trace stats of the MAF kernels (tools/time_maf.py): the file behind `maf_flow.roofline.profile` in the bench line
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_maf_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
python3 $R/tools/time_maf.py > "$OUT/time_maf.txt" 2>&1
cat "$OUT/time_maf.txt"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/tools/time_maf.py > "$OUT/under_rocprof.txt" 2> "$OUT/rocprof.log"
for f in $(find "$OUT" -name "*kernel_stats.csv"); do echo "== $f"; head -8 "$f" | cut -c1-200; cp "$f" "$OUT/maf_kernel_stats.csv"; done
