#!/bin/bash
# round 4, GPU batch 13: touch prefetch of the next frame in the backward kernel
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch13; mkdir -p $out
V=$PWD/gpurun_variants
timeout 300 python tools/r04/ab_inproc.py bwd base=$PWD/torchaudio-contrib_amd/libtac_amd.so touch=$V/libtac_bwd_touch.so noload=$V/libtac_bwd_noload.so 2>&1 | grep -v amdgpu.ids > $out/ab_bwd_touch.txt
cat $out/ab_bwd_touch.txt
