#!/bin/bash
# fused kernel: what moving the first exchange off the LDS would be worth — timing-only ablation with the exchange replaced by 32 MFMA
# transposes (wrong results: the back half of the transform is not written for the layout they produce)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python tools/r04/ab_inproc.py mel shipped=torchaudio-contrib_amd/libtac_amd.so mfma_x=gpurun_variants/libtac_mfmax.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch29_ab_mel_mfma_x.txt
cat gpurun_out/r04/batch29_ab_mel_mfma_x.txt
