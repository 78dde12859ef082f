#!/bin/bash
# round 4, GPU batch 6: 128-byte aligned row stores of stft_stream3_kernel
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch6; mkdir -p $out
V=$PWD/gpurun_variants
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_default.txt
for op in stft spec; do
  timeout 300 python tools/r04/ab_inproc.py $op noal=$V/libtac_s_noal.so al=$V/libtac_s_al.so al_plain=$V/libtac_s_al_plain.so al_nofft=$V/libtac_s_al_nofft.so r03=$V/libtac_r03.so 2>&1 | grep -v amdgpu.ids
done > $out/ab_stft_inproc.txt
TAC_ROTATE=4 timeout 200 python tools/time_steady.py stft spec mel 2>&1 | grep median > $out/time_steady.txt
cat $out/pytest_default.txt $out/ab_stft_inproc.txt $out/time_steady.txt
