#!/bin/bash
# hpss_tile8_kernel v3 (16-byte global accesses, stores in the B map): what each piece is worth
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -3 > gpurun_out/r04/batch25_tests.txt
V=gpurun_variants
for k in 31 17 9; do
python tools/r04/ab_other.py hpss:$k run4=$V/libtac_hp_run4.so v3=$V/libtac_hp_v3.so fill0=$V/libtac_hp_v3_fill0.so st0=$V/libtac_hp_v3_st0.so amap0=$V/libtac_hp_v3_amap0.so nt=$V/libtac_hp_v3_nt.so s96=$V/libtac_hp_v3_s96.so ns=$V/libtac_hp_v3_ns.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch25_ab_hpss.txt
cat gpurun_out/r04/batch25_tests.txt gpurun_out/r04/batch25_ab_hpss.txt
