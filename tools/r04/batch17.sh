#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch17; mkdir -p $out
timeout 300 python tools/r04/placement_scan.py spec 2>&1 | grep -v amdgpu.ids > $out/placement_spec.txt
timeout 300 python tools/r04/placement_scan.py stft 2>&1 | grep -v amdgpu.ids > $out/placement_stft.txt
cat $out/placement_spec.txt $out/placement_stft.txt
