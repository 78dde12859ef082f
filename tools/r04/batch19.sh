#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch19; mkdir -p $out
L=$PWD/torchaudio-contrib_amd/libtac_amd.so; V=$PWD/gpurun_variants
timeout 300 python tools/r04/ab_inproc.py mel classic=$L pieces5+p=$L pieces4+p=$V/libtac_pb4.so pieces6+p=$V/libtac_pb6.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_pieces.txt
cat $out/ab_mel_pieces.txt
