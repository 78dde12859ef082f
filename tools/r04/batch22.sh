#!/bin/bash
# phase_vocoder: fixed-point running phase, request depth, first-frame prefetch (same-process A/B) + its parity tests
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "phase_vocoder or g7 or float64 or opcheck or time_stretch" 2>&1 | tail -5 > gpurun_out/r04/batch22_tests.txt
V=gpurun_variants
for rate in 1.3 0.8 2.0; do
python tools/r04/ab_other.py pv:$rate r3=$V/libtac_pv_r3.so fix4=$V/libtac_pv_fix4.so fix8=$V/libtac_pv_fix8.so fix4p0=$V/libtac_pv_fix4p0.so fix8p0=$V/libtac_pv_fix8p0.so fix12p0=$V/libtac_pv_fix12p0.so fix16p0=$V/libtac_pv_fix16p0.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch22_ab_pv.txt
cat gpurun_out/r04/batch22_tests.txt gpurun_out/r04/batch22_ab_pv.txt
