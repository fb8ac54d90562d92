#!/bin/bash
# Profile bench.py on the MI355X box: kernel-trace stats, then PMC passes (each in its own run).
# usage: scripts/profile_round.sh TAG [extra bench.py flags, e.g. --dtype bf16]   (outputs under gpurun_out/prof_TAG/)
set -e
TAG=${1:-x}
shift || true
EXTRA="$*"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
B="python3 bench.py --no-cpu-baseline --no-augment --no-inference --no-bf16 $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $B --steps 6 --warmup 2 > "$OUT/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq" -o run -- $B --steps 2 --warmup 1 > "$OUT/pmc_sq.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -o run -- $B --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- $B --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1
python3 scripts/pmc_summarize.py "$OUT/pmc_summary.csv" --json "$OUT/pmc_traffic.json" --source "scripts/profile_round.sh $TAG $EXTRA" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write"
# keep only the small summaries for the merge back
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
ls -la "$OUT" "$OUT/stats"
