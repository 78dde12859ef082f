#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V=gpurun_variants
for k in 31 9; do
python tools/r04/ab_other.py hpss:$k coalesced=$V/libtac_hp_coal.so coalesced_nt=$V/libtac_hp_coal_nt.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch34_ab_hpss_nt.txt
cat gpurun_out/r04/batch34_ab_hpss_nt.txt
