#!/bin/bash
# Run on the GPU box: kernel trace of repeated spline training calls; prints the gaps between consecutive launches of a minibatch
set -u
OUT=$PWD/gpurun_out/trace_rows; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $R/tools/time_spline_train.py ${1:-50} > "$OUT/times.txt" 2> /dev/null )
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:30]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
gap = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if 'splr_' in n0 and 'splr_' in n1 or 'epoch_end' in n0 or 'epoch_end' in n1:
        gap.setdefault((n0.split('(')[0][-22:], n1.split('(')[0][-22:]), []).append(s1 - e0)
for k, v in gap.items():
    print('%-24s -> %-24s  n %5d  median gap %6.0f ns  p10 %6.0f  p90 %6.0f' % (k[0], k[1], len(v), st.median(v), sorted(v)[len(v) // 10], sorted(v)[len(v) * 9 // 10]))
dur = {}
for s0, e0, n0 in rows: dur.setdefault(n0.split('(')[0][-22:], []).append(e0 - s0)
for k, v in dur.items():
    if len(v) > 50: print('%-24s n %5d median duration %6.0f ns' % (k, len(v), st.median(v)))
PY
cat "$OUT/times.txt"
find "$OUT" -name "*.csv" -delete
