#!/bin/bash
# Run on the GPU box (via gpurun): the round's timing tables, one file each under gpurun_out/<tag>/, ready to be copied into
# profiles/<tag>/.  `tools/build_stamp.sh` must have been run in the build container first (tools/ab/lib_STAMP.so travels).
#   scripts/measure_round.sh [tag]
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python tools/time_quad.py                                             2>/dev/null > "$OUT/k4_forms_and_step_rules.txt"
NNEST_HIP_LIB=tools/ab/lib_STAMP.so python tools/time_quad.py stamp   2>/dev/null > "$OUT/k4_stamps.txt"
python tools/time_solo_wpg.py 50 2>/dev/null > "$OUT/k4_solo_wpg_sweep_d50.txt"
python tools/time_k4.py 50                                            2>/dev/null > "$OUT/k4_population_sweep_d50.txt"
python tools/time_k4.py 100                                           2>/dev/null > "$OUT/k4_population_sweep_d100.txt"
for occ in 2 3; do echo "# NNEST_MH_OCC=$occ (image form pinned to that build)"; NNEST_MH_OCC=$occ python tools/time_k4.py 50 image 2>/dev/null; done > "$OUT/k4_image_occupancy_d50.txt"
for occ in 1 2 3; do echo "# NNEST_MH_OCC=$occ (image form pinned to that build)"; NNEST_MH_OCC=$occ python tools/time_k4.py 100 image 2>/dev/null; done > "$OUT/k4_image_occupancy_d100.txt"
python tools/time_train.py                                            2>/dev/null > "$OUT/k5_epoch_times.txt"
python tools/time_train.py 100 8000                                   2>/dev/null >> "$OUT/k5_epoch_times.txt"
python tools/time_train.py 20 2000                                    2>/dev/null >> "$OUT/k5_epoch_times.txt"
NNEST_HIP_LIB=tools/ab/lib_STAMP.so python tools/time_train.py        2>/dev/null > "$OUT/k5_stamps.txt"
python tools/run_timing.py                                            2>/dev/null > "$OUT/run_timing_cfg2.txt"
python bench.py 2> "$OUT/bench_full.err" > "$OUT/bench_full.json"
python bench.py --config 5 --steps 5 --warmup 1 2> "$OUT/bench_cfg5.err" > "$OUT/bench_cfg5.json"
tail -n 3 "$OUT"/*.txt
cat "$OUT/bench_full.json"
