#!/bin/bash
# Registers / spills / LDS / occupancy of every kernel in one .hip file (cross-compiles, no GPU needed):
#   scripts/kernel_resources.sh leaffliction_amd/csrc/lf_wgrad_bf16.hip [filter]
F=$1; FILT=${2:-.}
EXTRA=""
case "$F" in *lf_augment*|*lf_geom*|*lf_filters*) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude $EXTRA -c "$F" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|Spill|Occupancy|LDS Size" \
 | sed -E 's/^.*remark: +//; s/ \[-Rpass.*//; s/^ +//' \
 | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' \
 | sed -E 's/Function Name: //; s/_ZN12_GLOBAL__N_1[0-9]+//; s/EvN2lf[A-Za-z0-9_]*//' | grep -E "$FILT" | cut -c1-200
