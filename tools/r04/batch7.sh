#!/bin/bash
# round 4, GPU batch 7: next frame requested a whole frame ahead (12-wave forms), with and without 128-byte aligned row stores
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch7; mkdir -p $out
V=$PWD/gpurun_variants
for op in stft spec; do
  timeout 300 python tools/r04/ab_inproc.py $op noal=$V/libtac_s_noal.so al=$V/libtac_s_al.so al_early=$V/libtac_s_al_early.so al_early12=$V/libtac_s_al_early12.so noal_early12=$V/libtac_s_noal_early12.so al12=$V/libtac_s_al12.so 2>&1 | grep -v amdgpu.ids
done > $out/ab_stft_inproc.txt
TAC_AMD_LIB=$V/libtac_s_al_early12.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 > $out/pytest_al_early12.txt
cat $out/ab_stft_inproc.txt $out/pytest_al_early12.txt
