#!/bin/bash
# Augmentation kernels on the MI355X box: kernel-trace stats of the whole table, then SQ counters of the
# north-star kernels one by one.  usage: scripts/profile_augment.sh TAG   (outputs under gpurun_out/prof_aug_TAG/)
set -e
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_aug_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/scripts/bench_augment.py" 4096 3 > "$OUT/stats.log" 2>&1
: > "$OUT/pmc_ops.txt"
for op in hist blur5 blur15 skew shear inclusive_mask; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d "$OUT/pmc_$op" -o p -- python3 "$ROOT/scripts/prof_one_op.py" $op > "$OUT/pmc_$op.log" 2>&1
  echo "== $op (4096 x 224x224x3; counters summed over the chip, per launch)" >> "$OUT/pmc_ops.txt"
  python3 "$ROOT/scripts/pmc_db.py" "$OUT/pmc_$op/p_results.db" >> "$OUT/pmc_ops.txt"
  rm -rf "$OUT/pmc_$op"
done
find "$OUT" -name "*kernel_trace.csv" -delete
cat "$OUT/pmc_ops.txt" | cut -c1-150
