#!/bin/bash
# hpss: persistent workgroups that request the next tile before storing the current one (2 / 3 / 4 per CU) vs one tile per workgroup
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -3 > gpurun_out/r04/batch36_tests.txt
TAC_FUZZ_CASES=300 TAC_FUZZ_SEED=9 python -m pytest tests/test_gpu_fuzz.py -x -q -k hpss 2>&1 | tail -3 >> gpurun_out/r04/batch36_tests.txt
V=gpurun_variants
for k in 31 17 9; do
python tools/r04/ab_other.py hpss:$k one_tile=$V/libtac_hp_np.so persist3=$V/libtac_hp_p3.so persist2=$V/libtac_hp_p2.so persist4=$V/libtac_hp_p4.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch36_ab_hpss_persist.txt
cat gpurun_out/r04/batch36_tests.txt gpurun_out/r04/batch36_ab_hpss_persist.txt
