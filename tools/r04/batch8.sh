#!/bin/bash
# round 4, GPU batch 8: LDS trims of melspec_stream3_kernel — R2C twiddles / slot-0 weights in registers, row written as write2st64 pairs
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch8; mkdir -p $out
V=$PWD/gpurun_variants
timeout 300 python tools/r04/ab_inproc.py mel base=$PWD/torchaudio-contrib_amd/libtac_amd.so ptwr=$V/libtac_ptwr.so st64=$V/libtac_st64.so ptwrst64=$V/libtac_ptwrst64.so w0r=$V/libtac_w0r.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_inproc.txt
TAC_AB_N=400 timeout 300 python tools/r04/ab_inproc.py mel base=$PWD/torchaudio-contrib_amd/libtac_amd.so ptwrst64=$V/libtac_ptwrst64.so 2>&1 | grep -v amdgpu.ids >> $out/ab_mel_inproc.txt
TAC_AMD_LIB=$V/libtac_ptwrst64.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mel or g1 or g2 or g9 or cfg" 2>&1 | tail -3 > $out/pytest_ptwrst64.txt
cat $out/ab_mel_inproc.txt $out/pytest_ptwrst64.txt
