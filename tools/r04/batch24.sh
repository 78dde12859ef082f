#!/bin/bash
# hpss_tile8_kernel: tile fill, LDS row stride, occupancy hint, A map (column pairs vs single columns)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -3 > gpurun_out/r04/batch24_tests.txt
V=gpurun_variants
for k in 31 9; do
python tools/r04/ab_other.py hpss:$k run4=$V/libtac_hp_run4.so fill0=$V/libtac_hp_fill0.so fill1=$V/libtac_hp_fill1.so s96=$V/libtac_hp_s96.so s104=$V/libtac_hp_s104.so occ4=$V/libtac_hp_occ4.so amap1=$V/libtac_hp_amap1.so amap1s96=$V/libtac_hp_amap1_s96.so amap1ns=$V/libtac_hp_amap1_ns.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch24_ab_hpss.txt
cat gpurun_out/r04/batch24_tests.txt gpurun_out/r04/batch24_ab_hpss.txt
