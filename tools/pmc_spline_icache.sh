#!/bin/bash
# Developer diagnostic (GPU box): instruction-cache counters of the spline gradient kernel
export TMPDIR=/tmp; R=$PWD; OUT=/tmp/splt_pmc2; rm -rf $OUT
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQ_WAIT_IFETCH|SQ_INST_LEVEL|SQC_" | head -40
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/time_spline_train.py ${1:-50} > /dev/null 2>$OUT.log || tail -5 $OUT.log
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/splt_pmc2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spl_grad' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0][-30:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print('    %-28s median %14.1f  max %14.1f (n=%d)' % (c, v[len(v) // 2], v[-1], len(v)))
PY
