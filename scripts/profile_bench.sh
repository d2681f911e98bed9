#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes (separate runs, --kernel-trace only beside --pmc)
# of the default bench command.  Writes the summaries to gpurun_out/prof_<tag>/ ready to be copied into profiles/<tag>/:
#   bench_kernel_stats.csv   per-kernel durations (the K4 average must agree with bench.py's roofline.kernel_ms)
#   bench_pmc_summary.txt    FETCH_SIZE / WRITE_SIZE / SQ counters, mean per K4 dispatch
#   bench_pmc.json           {"traffic_bytes_per_launch": ...} read back by bench.py (roofline.traffic)
#   bench_under_rocprof.json the bench line printed under the profiler
set -u
TAG=${1:-r06}
KERN=${2:-mh_kernel_solo<2, false, 0, 4>}   # the kernel the default bench command runs (roofline.kernel)
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 50 --warmup 50 --bare"   # (bench.py's defaults)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $BENCH > /dev/null 2> "$OUT/pmc_sq.log"
cd "$OUT"
f=$(find trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" > bench_kernel_stats.csv
python3 - "$TAG" "$KERN" <<'PY'
import csv, glob, collections, json, sys
tag, kern = sys.argv[1], sys.argv[2]
vals = {}
n = 0
for d in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
    for f in glob.glob('%s/**/*counter_collection.csv' % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for c, v in agg.items():
            vals[c] = sum(v) / len(v)
            n = len(v)
with open('bench_pmc_summary.txt', 'w') as o:
    o.write('# rocprofv3 PMC passes of `python bench.py --steps 50 --warmup 50 --bare` (scripts/profile_bench.sh %s);\n'
            '# mean per dispatch of %s (%d dispatches); FETCH_SIZE / WRITE_SIZE raw counter units are KiB; SQ cycle counters count '
            '4 clocks\n' % (tag, kern, n))
    for c in sorted(vals):
        o.write('%-18s %14.2f\n' % (c, vals[c]))
if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
    json.dump({'traffic_bytes_per_launch': int((vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024), 'fetch_kib': vals['FETCH_SIZE'],
               'write_kib': vals['WRITE_SIZE'], 'kernel': kern.split('<')[0], 'source': 'scripts/profile_bench.sh ' + tag,
               'note': 'raw FETCH_SIZE + WRITE_SIZE (one dword per lane accesses: reads ~0.8x, writes exact, DESIGN.md 3)'},
              open('bench_pmc.json', 'w'))
print(open('bench_pmc_summary.txt').read())
PY
cat bench_kernel_stats.csv
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
