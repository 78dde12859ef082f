#!/bin/bash
# hpss: wide tiles (32 x 128, 16 x 256: result rows leave as 512-byte / 1 KB segments) vs 64 x 64
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V=gpurun_variants
for v in 32x128 16x256; do
  TAC_AMD_LIB=$PWD/$V/libtac_hp_$v.so python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -2
  TAC_AMD_LIB=$PWD/$V/libtac_hp_$v.so TAC_FUZZ_CASES=200 TAC_FUZZ_SEED=13 python -m pytest tests/test_gpu_fuzz.py -x -q -k hpss 2>&1 | tail -2
done > gpurun_out/r04/batch39_tests.txt
for k in 31 9; do
python tools/r04/ab_other.py hpss:$k t64x64=$V/libtac_hp_64.so t32x128=$V/libtac_hp_32x128.so t16x256=$V/libtac_hp_16x256.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch39_ab_hpss_wide.txt
cat gpurun_out/r04/batch39_tests.txt gpurun_out/r04/batch39_ab_hpss_wide.txt
