#!/bin/bash
# round 4, GPU batch 3: RCCL tests again; same-process A/B of the row-store policy / pacing and of the LDS knobs; stage stamps
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch3; mkdir -p $out
V=$PWD/gpurun_variants
timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -x -q 2>&1 | tail -8 > $out/pytest_rccl.txt
for op in stft spec; do
  timeout 300 python tools/r04/ab_inproc.py $op r03=$V/libtac_r03.so nt=$V/libtac_s_nt.so plain=$V/libtac_s_plain.so nt_vm0=$V/libtac_s_nt_vm0.so plain_vm0=$V/libtac_s_plain_vm0.so 2>&1 | grep -v amdgpu.ids
done > $out/ab_stft_inproc.txt
timeout 300 python tools/r04/ab_inproc.py mel r03=$V/libtac_r03.so lean=$PWD/torchaudio-contrib_amd/libtac_amd.so b64=$V/libtac_b64.so w0=$V/libtac_w0.so b64w0=$V/libtac_b64w0.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_inproc.txt
TAC_AMD_LIB=$V/libtac_stamps.so timeout 200 python tools/r04/s3_stamps.py 2>&1 | grep -v amdgpu.ids > $out/s3_stamps.txt
cat $out/pytest_rccl.txt $out/ab_stft_inproc.txt $out/ab_mel_inproc.txt $out/s3_stamps.txt
