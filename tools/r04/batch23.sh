#!/bin/bash
# hpss: runs of eight windows (median_run8) vs runs of four, with / without stores; phase_vocoder access-pattern-only ablation
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -5 > gpurun_out/r04/batch23_tests.txt
V=gpurun_variants
for k in 31 17 9; do
python tools/r04/ab_other.py hpss:$k run4=$V/libtac_hp_run4.so run8=$V/libtac_hp_run8.so run4ns=$V/libtac_hp_run4_nostore.so run8ns=$V/libtac_hp_run8_nostore.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch23_ab_hpss.txt
python tools/r04/ab_other.py pv:1.3 fix4=$V/libtac_pv_fix4.so copy=$V/libtac_pv_copy.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch23_ab_pv_copy.txt
cat gpurun_out/r04/batch23_tests.txt gpurun_out/r04/batch23_ab_hpss.txt gpurun_out/r04/batch23_ab_pv_copy.txt
