#!/bin/bash
# melspec_backward_ring3_kernel: nontemporal stores of the finished gradient samples / nontemporal loads of the mel-gradient row
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V=gpurun_variants
python tools/r04/ab_inproc.py bwd shipped=torchaudio-contrib_amd/libtac_amd.so nt_st=$V/libtac_br_nt1.so nt_ld=$V/libtac_br_nt2.so nt_both=$V/libtac_br_nt3.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch30_ab_bwd_nt.txt
cat gpurun_out/r04/batch30_ab_bwd_nt.txt
