#!/bin/bash
# fused kernel: bounds of a producer / consumer split — no weight reads (consumers hold the bank), no window / pass-1 twiddle reads
# (producers, freed of the contraction's registers, could hold both tables); timing-only ablations
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V=gpurun_variants
python tools/r04/ab_inproc.py mel shipped=torchaudio-contrib_amd/libtac_amd.so no_weights=$V/libtac_noweights.so no_tables=$V/libtac_notables.so neither=$V/libtac_notables_noweights.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch41_ab_mel_bounds.txt
cat gpurun_out/r04/batch41_ab_mel_bounds.txt
