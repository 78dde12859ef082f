#!/bin/bash
# round 4, GPU batch 5: memory-system ceilings on the same box as the FFT-less ablation; radix-16 factor folding; early R2C-twiddle reads
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch5; mkdir -p $out
V=$PWD/gpurun_variants
timeout 120 tools/ubench/build/hbm_rate > $out/hbm_rate.txt 2>&1
timeout 200 tools/ubench/build/row_store_rate > $out/row_store_rate.txt 2>&1
for op in stft spec; do
  timeout 300 python tools/r04/ab_inproc.py $op nt=$V/libtac_s_nt.so fold=$V/libtac_s_fold.so nofft=$V/libtac_s_nofft.so 2>&1 | grep -v amdgpu.ids
done > $out/ab_stft_inproc.txt
timeout 300 python tools/r04/ab_inproc.py mel b64=$PWD/torchaudio-contrib_amd/libtac_amd.so nob64=$V/libtac_nob64.so fold=$V/libtac_fold.so ptwe=$V/libtac_ptwe.so foldptwe=$V/libtac_foldptwe.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_inproc.txt
TAC_AMD_LIB=$V/libtac_foldptwe.so timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_foldptwe.txt
TAC_AMD_LIB=$V/libtac_s_fold.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stft or spectrogram or g1 or g4" 2>&1 | tail -3 > $out/pytest_s_fold.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_default.txt
cat $out/hbm_rate.txt $out/row_store_rate.txt $out/ab_stft_inproc.txt $out/ab_mel_inproc.txt $out/pytest_*.txt
