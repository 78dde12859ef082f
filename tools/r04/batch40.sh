#!/bin/bash
# fused kernel: upper bound of what weights-in-registers consumer waves could save — timing-only ablation without the 18 weight reads per frame
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python tools/r04/ab_inproc.py mel shipped=torchaudio-contrib_amd/libtac_amd.so no_weight_reads=gpurun_variants/libtac_noweights.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch40_ab_mel_noweights.txt
cat gpurun_out/r04/batch40_ab_mel_noweights.txt
