#!/bin/bash
# hpss: stores through a per-wave LDS transpose (whole 256-byte row segments per store instruction) vs 16-byte pieces 32 bytes apart
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -3 > gpurun_out/r04/batch33_tests.txt
TAC_FUZZ_CASES=200 TAC_FUZZ_SEED=7 python -m pytest tests/test_gpu_fuzz.py -x -q -k hpss 2>&1 | tail -3 >> gpurun_out/r04/batch33_tests.txt
V=gpurun_variants
for k in 31 9 5x9; do
python tools/r04/ab_other.py hpss:$k pieces=$V/libtac_hp_nocoal.so coalesced=$V/libtac_hp_coal.so nostore=$V/libtac_hp_coal_ns.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch33_ab_hpss_coalesce.txt
cat gpurun_out/r04/batch33_tests.txt gpurun_out/r04/batch33_ab_hpss_coalesce.txt
