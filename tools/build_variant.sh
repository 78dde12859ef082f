#!/bin/bash
# Build an A/B variant of libtac_amd.so:  tools/build_variant.sh NAME "-DTAC_X=1 ..." [file.hip ...]
# Recompiles the listed sources (default: melspec_sparse.hip) with the extra flags and links them with the default
# objects into gpurun_variants/libtac_NAME.so (git-ignored, travels with gpurun; select with TAC_AMD_LIB=...).
set -e
name=$1; flags=$2; shift 2
files=${@:-melspec_sparse.hip}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/torchaudio-contrib_amd/csrc
out=$root/gpurun_variants
mkdir -p $out/build_$name
[ -n "$TAC_NO_MAKE" ] || make -s -C $src >/dev/null
objs=""
for f in $src/build/*.o; do
  b=$(basename $f .o)
  if [[ " $files " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable $flags -c $src/$b.hip -o $out/build_$name/$b.o
    objs="$objs $out/build_$name/$b.o"
  else
    objs="$objs $f"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libtac_$name.so $objs
echo built $out/libtac_$name.so
