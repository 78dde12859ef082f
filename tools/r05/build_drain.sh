#!/bin/bash
# Build an A/B library with the wave-specialised-store form of the fft_length-2048 STFT rows (tools/ablation/stft_drain3.hpp):
#   tools/r05/build_drain.sh NAME TWc DWc TWr DWr [extra -D flags]
# TWc + DWc transform / drain waves for the complex rows, TWr + DWr for the real rows (sum 12 or 16).  Flags of interest:
#   -DTAC_S3_DRAIN_INPLACE=0|1 (LDS ring | rows in place + tail buffers)   -DTAC_S3_DRAIN_PRIO=3   -DTAC_S3_DRAIN_PIPE=0|1
#   -DTAC_S3_DRAIN_LATEPUB=1   -DTAC_S3_DRAIN_ABL=1|2 (timing only)   -DTAC_S3_DRAIN_STAMPS=1 (tools/r05/drain_stamps.py)
# The header and its launch hook live outside csrc/: they are copied / patched in for the build and removed again.
set -e
name=$1; twc=$2; dwc=$3; twr=$4; dwr=$5; shift 5
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/torchaudio-contrib_amd/csrc
make -s -C $src >/dev/null
base=$(mktemp -d); tmp=$base/pkg/csrc                     # (host_common.hpp includes "../../include/tac_amd.h")
mkdir -p $tmp $base/include
cp $src/*.hip $src/*.hpp $tmp/ && cp $root/tools/ablation/stft_drain3.hpp $tmp/ && cp $root/include/tac_amd.h $base/include/
(cd $tmp && patch -s -p3 < $root/tools/ablation/stft_drain3_hook_r05.patch)
out=$root/gpurun_variants; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable \
  -DTAC_S3_DRAIN=1 -DTAC_S3_DRAIN_TW_C=$twc -DTAC_S3_DRAIN_DW_C=$dwc -DTAC_S3_DRAIN_TW_R=$twr -DTAC_S3_DRAIN_DW_R=$dwr "$@" -c $tmp/stft_kernels.hip -o $tmp/stft_kernels.o
objs=""
for f in $src/build/*.o; do [ "$(basename $f)" = stft_kernels.o ] && objs="$objs $tmp/stft_kernels.o" || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libtac_$name.so $objs
rm -rf $base
echo built $out/libtac_$name.so
