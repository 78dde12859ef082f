#!/bin/bash
# tools/build_one.sh <source.hip> [extra flags]: compile ONE source of csrc/ with the Makefile's flags, print the register report of
# the kernels whose name matches $KERNEL (default: all), keep the assembly under csrc/build/
set -e
cd "$(dirname "$0")/../torchaudio-contrib_amd/csrc"
src=$1; shift
mkdir -p build/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj "$@" -c $src -o build/asm/${src%.hip}.o > build/${src%.hip}.remarks 2>&1 || { grep -B2 -A8 "error" build/${src%.hip}.remarks | head -60; exit 1; }
grep -A10 "Function Name: .*${KERNEL:-}" build/${src%.hip}.remarks | grep -E "Name|VGPRs:|VGPRs Spill|Occupancy" | sed -e 's/\[-Rpass.*//' -e 's/remark: [^ ]* *//'
