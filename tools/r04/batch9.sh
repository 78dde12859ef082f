#!/bin/bash
# round 4, GPU batch 9: new defaults through the whole GPU suite; backward kernel with single b64 read-backs (same-process A/B)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch9; mkdir -p $out
V=$PWD/gpurun_variants
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_default.txt
timeout 300 python tools/r04/ab_inproc.py bwd b64=$PWD/torchaudio-contrib_amd/libtac_amd.so read2=$V/libtac_bwd_r2.so r03=$V/libtac_r03.so 2>&1 | grep -v amdgpu.ids > $out/ab_bwd_inproc.txt
timeout 300 python tools/r04/ab_inproc.py mel new=$PWD/torchaudio-contrib_amd/libtac_amd.so r03=$V/libtac_r03.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_inproc.txt
for op in stft spec; do timeout 300 python tools/r04/ab_inproc.py $op new=$PWD/torchaudio-contrib_amd/libtac_amd.so r03=$V/libtac_r03.so 2>&1 | grep -v amdgpu.ids; done > $out/ab_stft_inproc.txt
cat $out/pytest_default.txt $out/ab_bwd_inproc.txt $out/ab_mel_inproc.txt $out/ab_stft_inproc.txt
