#!/bin/bash
# Developer diagnostic (GPU box): instruction-cache and issue counters of the spline proposal kernel, pair form against team form
#   bash tools/pmc_spline_mh.sh > gpurun_out/pmc_spline_mh.txt
export TMPDIR=/tmp; R=$PWD
cd /tmp
for form in pair team; do
    export NNEST_SPLINE_MH_FORM=$form
    OUT=/tmp/splmh_$form; rm -rf $OUT
    rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/time_spline.py 50 1000 > /dev/null 2>$OUT.a.log || tail -5 $OUT.a.log
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/b -- python3 $R/tools/time_spline.py 50 1000 > /dev/null 2>$OUT.b.log || tail -5 $OUT.b.log
    python3 - $form <<'PY'
import csv, glob, collections, sys
form = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/splmh_%s/**/*counter_collection.csv' % form, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spline_mh_kernel' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print('NNEST_SPLINE_MH_FORM=%s  %s' % (form, k))
    for c, v in sorted(d.items()):
        v = sorted(v)
        print('    %-28s median %16.1f  (n=%d)' % (c, v[len(v) // 2], len(v)))
PY
done
