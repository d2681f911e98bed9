#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + SQ counters of the training kernels (tools/time_train.py:
# the eight-CU kernel and the one-CU kernel at config 2).  Summaries to gpurun_out/prof_train_<tag>/ for profiles/<tag>/.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_train_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/tools/time_train.py 50 1000 > "$OUT/train_epoch_times_under_rocprof.txt" 2> "$OUT/trace.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/pmc" -- python3 $R/tools/time_train.py 50 1000 > /dev/null 2> "$OUT/pmc.log"
cd "$OUT"
f=$(find trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" > train_kernel_stats.csv
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'train_kernel' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0][-44:]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('train_pmc_summary.txt', 'w') as o:
    o.write('# rocprofv3 --pmc over tools/time_train.py 50 1000 (scripts/profile_train.sh %s): mean over the 4 dispatches of each kernel (warm-up and timed, 2 to 40 epochs of 9 minibatches);\n'
            '# SQ cycle counters count 4 clocks\n' % sys.argv[1])
    for k, d in sorted(agg.items()):
        o.write(k + '\n')
        for c, v in sorted(d.items()):
            o.write('    %-18s %16.1f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open('train_pmc_summary.txt').read())
PY
cat train_kernel_stats.csv | cut -c1-160
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
