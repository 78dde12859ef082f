#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
L=torchaudio-contrib_amd/libtac_amd.so
V=gpurun_variants
for op in stft spec; do
timeout 300 env TAC_AB_N=300 python tools/r04/ab_inproc.py $op ring=$L aligned=$V/libtac_al.so 2>&1 | grep -v "amdgpu.ids"
done | tee gpurun_out/r05/batch24_ab_ring_aligned.txt
TAC_AMD_LIB=$V/libtac_al.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stft or spectrogram or g1 or g4 or cfg2 or layout or cfg3 or hop_ring" 2>&1 | tail -4 | tee -a gpurun_out/r05/batch24_ab_ring_aligned.txt
