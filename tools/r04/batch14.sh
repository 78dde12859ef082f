#!/bin/bash
# round 4, GPU batch 14: XOR-swizzled first exchange + dense partner exchange (no LDS bank conflicts on either)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch14; mkdir -p $out
V=$PWD/gpurun_variants
timeout 300 python tools/r04/ab_inproc.py mel base=$PWD/torchaudio-contrib_amd/libtac_amd.so swz=$V/libtac_swz.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_swz.txt
for op in stft spec; do timeout 300 python tools/r04/ab_inproc.py $op base=$PWD/torchaudio-contrib_amd/libtac_amd.so swz=$V/libtac_s_swz.so 2>&1 | grep -v amdgpu.ids; done > $out/ab_stft_swz.txt
TAC_AMD_LIB=$V/libtac_swz.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 > $out/pytest_swz.txt
TAC_AMD_LIB=$V/libtac_s_swz.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 > $out/pytest_s_swz.txt
cat $out/ab_mel_swz.txt $out/ab_stft_swz.txt $out/pytest_swz.txt $out/pytest_s_swz.txt
