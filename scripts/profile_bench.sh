#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench command.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-saturation --no-spline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $BENCH > /dev/null 2> "$OUT/pmc_sq.log"
cd "$OUT"
find . -name "*.csv" | head -50
for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -12 "$f"; done
python3 - <<'PY'
import csv, glob, collections
for tag in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
    for f in glob.glob('%s/**/*counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            if 'mh_kernel' in k or 'flow_pass' in k:
                print(tag, k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
