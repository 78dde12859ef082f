#!/bin/bash
# phase_vocoder: the running phase as a unit phasor advanced by complex products (no atan2 / sin / cos) vs the fixed-point turn fraction
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
(python -m pytest tests -m gpu -x -q -k "phase_vocoder or g7 or float64 or opcheck or time_stretch" 2>&1 | tail -4
TAC_FUZZ_CASES=300 TAC_FUZZ_SEED=5 python -m pytest tests/test_gpu_fuzz.py -x -q -k phase_vocoder 2>&1 | tail -4) > gpurun_out/r04/batch32_tests.txt
V=gpurun_variants
for rate in 1.3 0.8 2.0; do
python tools/r04/ab_other.py pv:$rate fixed=$V/libtac_pv_fixed.so phasor=$V/libtac_pv_phasor.so copy=$V/libtac_pv_copy.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch32_ab_pv_phasor.txt
cat gpurun_out/r04/batch32_tests.txt gpurun_out/r04/batch32_ab_pv_phasor.txt
