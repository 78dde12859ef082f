#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch21; mkdir -p $out
L=$PWD/torchaudio-contrib_amd/libtac_amd.so
timeout 300 python tools/r04/ab_inproc.py mel classic=$L pieces+p=$L 2>&1 | grep -v amdgpu.ids > $out/ab_mel_pieces.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $out/pytest_default.txt
cat $out/ab_mel_pieces.txt $out/pytest_default.txt
