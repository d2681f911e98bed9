#!/bin/bash
# rocprofv3 PMC pass over the training kernels (tools/time_train.py, tools/time_spline_train.py): mean per dispatch
set -u
TAG=${1:-r01e}
OUT=$PWD/gpurun_out/pmc_train_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/nvp" -- python3 $R/tools/time_train.py > /dev/null 2> "$OUT/nvp.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/spline" -- python3 $R/tools/time_spline_train.py 50 > /dev/null 2> "$OUT/spline.log"
cd "$OUT"
python3 - <<'PY'
import csv, glob, collections
for tag in ('nvp', 'spline'):
    for f in glob.glob('%s/**/*counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in sorted(agg.items()):
            if 'train_kernel' in k or 'spl_' in k:
                print(tag, k)
                for c, v in sorted(d.items()):
                    print('    %-18s %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
