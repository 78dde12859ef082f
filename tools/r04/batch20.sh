#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch20; mkdir -p $out
L=$PWD/torchaudio-contrib_amd/libtac_amd.so; V=$PWD/gpurun_variants
timeout 300 python tools/r04/ab_inproc.py mel classic=$L pieces4+p=$V/libtac_pb4.so noadds+p=$V/libtac_pa1.so plainst+p=$V/libtac_pa2.so both+p=$V/libtac_pa3.so 2>&1 | grep -v amdgpu.ids > $out/ab_mel_pieces_abl.txt
cat $out/ab_mel_pieces_abl.txt
