#!/bin/bash
# hpss: time against the number of result arrays actually stored (timing-only ablation)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V=gpurun_variants
for k in 31 9; do
python tools/r04/ab_other.py hpss:$k four=torchaudio-contrib_amd/libtac_amd.so three=$V/libtac_hp_a3.so two=$V/libtac_hp_a2.so one=$V/libtac_hp_a1.so none=$V/libtac_hp_a0.so 2>&1 | grep -v "amdgpu.ids\|^check"
done > gpurun_out/r04/batch35_ab_hpss_arrays.txt
cat gpurun_out/r04/batch35_ab_hpss_arrays.txt
