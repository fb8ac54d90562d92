#!/bin/bash
# Profile the bf16 inference forward pass (batch 1024): kernel-trace stats, then PMC passes (own runs).
# usage: scripts/profile_infer.sh TAG   (outputs under gpurun_out/prof_TAG/)
set -e
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
B="python3 scripts/infer_bf16.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $B > "$OUT/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq" -o run -- $B > "$OUT/pmc_sq.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -o run -- $B > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- $B > "$OUT/pmc_write.log" 2>&1
python3 scripts/pmc_summarize.py "$OUT/pmc_summary.csv" --json "$OUT/pmc_traffic.json" --source "scripts/profile_infer.sh $TAG" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write"
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
