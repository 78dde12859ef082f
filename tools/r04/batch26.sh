#!/bin/bash
# hpss with unequal / small widths: two tile launches (one halo axis each) vs round 3's one-thread-per-element kernel
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -3 > gpurun_out/r04/batch26_tests.txt
V=gpurun_variants
for k in 5x9 31x17 3x3 31; do
python tools/r04/ab_other.py hpss:$k elem=$V/libtac_hp_elem.so twopass=$V/libtac_hp_now.so run4=$V/libtac_hp_run4.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch26_ab_hpss.txt
cat gpurun_out/r04/batch26_tests.txt gpurun_out/r04/batch26_ab_hpss.txt
