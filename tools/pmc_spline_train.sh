#!/bin/bash
# Developer diagnostic (GPU box): SQ counters of the spline training kernels (tools/time_spline_train.py)
export TMPDIR=/tmp; R=$PWD; OUT=/tmp/splt_pmc; rm -rf $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/time_spline_train.py ${1:-50} > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -- python3 $R/tools/time_spline_train.py ${1:-50} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/splt_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spl_grad' in r['Kernel_Name'] or 'spl_w3' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0][-30:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print('    %-22s median %14.1f  max %14.1f (n=%d)' % (c, v[len(v) // 2], v[-1], len(v)))
PY
