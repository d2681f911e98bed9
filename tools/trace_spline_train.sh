#!/bin/bash
# Developer diagnostic (GPU box): per-kernel average durations of spline training under rocprofv3, for the library in $NNEST_HIP_LIB
export TMPDIR=/tmp; R=$PWD; TAG=${1:-x}
cd /tmp; rm -rf /tmp/splt_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/splt_$TAG -- python3 $R/tools/time_spline_train.py ${2:-50} > /dev/null 2>&1
f=$(find /tmp/splt_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$TAG" <<'PY'
import csv, sys
print('==', sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'spl_' in r['Name'] and int(r['Calls']) > 100:
        print('  %-34s calls %5s  avg %8.1f us  min %7.1f' % (r['Name'].replace('void ', '').replace('nnest::', '')[:34], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
