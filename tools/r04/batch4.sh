#!/bin/bash
# round 4, GPU batch 4: lane-to-lane R2C partners (ds_bpermute) + single b64 read-backs + division-free request; FFT-less ablation of the STFT rows
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch4; mkdir -p $out
V=$PWD/gpurun_variants
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "placement" 2>&1 | tail -3 > $out/pytest_pack.txt
for v in b64bpermnd; do
  TAC_AMD_LIB=$V/libtac_$v.so timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_$v.txt
done
TAC_AMD_LIB=$V/libtac_s_bp.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stft or spectrogram or g1 or g4" 2>&1 | tail -3 > $out/pytest_s_bp.txt
timeout 300 python tools/r04/ab_inproc.py mel lean=$PWD/torchaudio-contrib_amd/libtac_amd.so b64=$V/libtac_b64.so bperm=$V/libtac_bperm.so b64bperm=$V/libtac_b64bperm.so b64bpermnd=$V/libtac_b64bpermnd.so r03=$V/libtac_r03.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_inproc.txt
for op in stft spec; do
  timeout 300 python tools/r04/ab_inproc.py $op nt=$V/libtac_s_nt.so bp=$V/libtac_s_bp.so nofft=$V/libtac_s_nofft.so nofft_plain=$V/libtac_s_nofft_plain.so 2>&1 | grep -v amdgpu.ids
done > $out/ab_stft_inproc.txt
TAC_AMD_LIB=$V/libtac_stamps2.so timeout 200 python tools/r04/s3_stamps.py 2>&1 | grep -v amdgpu.ids > $out/s3_stamps2.txt
cat $out/pytest_*.txt $out/ab_mel_inproc.txt $out/ab_stft_inproc.txt $out/s3_stamps2.txt
