#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
L=torchaudio-contrib_amd/libtac_amd.so
V=gpurun_variants
for op in stft spec; do
timeout 300 env TAC_AB_N=300 python tools/r04/ab_inproc.py $op base=$L ring8=$V/libtac_ring8.so ring9=$V/libtac_ring9.so ring10=$V/libtac_ring10.so ring11=$V/libtac_ring11.so ring12=$V/libtac_ring12.so newhop=$V/libtac_newhop.so 2>&1 | grep -v "amdgpu.ids\|^check"
done | tee gpurun_out/r05/batch13_ab_ring_pf.txt
