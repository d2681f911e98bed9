#!/bin/bash
# PMC pass over a chip-filling population (developer diagnostic): bash scripts/pmc_sat.sh <walkers> <steps>
set -u
W=${1:-32768}; S=${2:-50}
OUT=$PWD/gpurun_out/pmc_sat; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-saturation --no-spline --walkers $W --mcmc-steps $S"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/a" -- $BENCH > "$OUT/bench.json" 2> "$OUT/a.log"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/b" -- $BENCH > /dev/null 2> "$OUT/b.log"
cd "$OUT"
python3 - <<'PY'
import csv, glob, collections
for tag in ('a', 'b'):
    for f in glob.glob('%s/**/*counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            if 'mh_kernel' in k:
                print(tag, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
    for f in glob.glob('%s/**/*kernel_trace.csv' % tag, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'mh_kernel' in r['Kernel_Name']]
        if rows:
            d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
            print(tag, 'mh durations ns', d, 'VGPR', rows[0].get('VGPR_Count'), 'accum', rows[0].get('Accum_VGPR_Count'), 'LDS', rows[0].get('LDS_Block_Size'), 'grid', rows[0].get('Grid_Size'), 'wg', rows[0].get('Workgroup_Size'))
PY
tail -3 "$OUT/a.log" "$OUT/b.log" | head -20
