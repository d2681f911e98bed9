#!/bin/bash
# PMC counters of the spline kernels (tools/time_spline.py) on the GPU box
set -u
OUT=$PWD/gpurun_out/pmc_spline; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/sq" -- python3 $R/tools/time_spline.py > /dev/null 2> "$OUT/sq.log"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/sq2" -- python3 $R/tools/time_spline.py > /dev/null 2> "$OUT/sq2.log"
cd "$OUT"
python3 - <<'PY'
import csv, glob, collections
for tag in ('sq', 'sq2'):
    for f in glob.glob('%s/**/*counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            if 'spline_mh' in k or 'spl_grad' in k or 'splr_' in k or 'spl_update' in k:
                print(tag, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'n', len(next(iter(d.values()))))
PY
tail -3 "$OUT/sq2.log"
