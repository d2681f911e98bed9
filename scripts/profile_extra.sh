#!/bin/bash
# rocprofv3 kernel-trace stats of (a) the training kernel at the BASELINE configs, (b) K4 at a chip-filling population
set -u
TAG=${1:-r01b}
OUT=$PWD/gpurun_out/prof_extra_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train" -- python3 $R/tools/time_train.py > "$OUT/train_stdout.txt" 2> "$OUT/train.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sat" -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-saturation --no-spline --walkers 131072 --mcmc-steps 25 > "$OUT/sat_stdout.json" 2> "$OUT/sat.log"
for f in $(find "$OUT" -name "*kernel_stats.csv"); do echo "== $f"; head -5 "$f" | cut -c1-200; done
cat "$OUT/train_stdout.txt"
