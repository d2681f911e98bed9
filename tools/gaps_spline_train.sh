#!/bin/bash
# Developer diagnostic (GPU box): start-to-start / end-to-start gaps between the kernels of spline training (rocprofv3 kernel trace)
export TMPDIR=/tmp; R=$PWD
cd /tmp; rm -rf /tmp/splt_gap
rocprofv3 --kernel-trace --output-format csv -d /tmp/splt_gap -- python3 $R/tools/time_spline_train.py 50 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/splt_gap/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'spl_' in r['Kernel_Name']]
n = len(rows)
seg = rows[n // 2: n // 2 + 400]
import collections
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(seg, seg[1:]):
    ka = a['Kernel_Name'].split('(')[0].split('::')[-1][:18]; kb = b['Kernel_Name'].split('(')[0].split('::')[-1][:18]
    gap[ka + ' -> ' + kb].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
    dur[ka].append((int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3)
for k, v in gap.items(): print('gap %-44s mean %6.2f us  min %6.2f (n=%d)' % (k, sum(v) / len(v), min(v), len(v)))
for k, v in dur.items(): print('dur %-44s mean %6.2f us  min %6.2f' % (k, sum(v) / len(v), min(v)))
t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
print('span %.1f us over %d kernels' % ((t1 - t0) / 1e3, len(seg)))
PY
