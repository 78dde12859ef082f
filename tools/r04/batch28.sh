#!/bin/bash
# stft_stream3_kernel: full factorial of waves {12,16} x 128-byte aligned stores x early request x bpermute partners (+ plain stores),
# complex rows and power rows, one process each (same-process A/B; first build = the shipped configuration per op)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V=gpurun_variants
args=""
for n in w12a0e1b0 w12a0e0b0 w12a0e0b1 w12a0e1b1 w12a1e0b0 w12a1e0b1 w12a1e1b0 w12a1e1b1 w16a0e0b0 w16a0e0b1 w16a1e0b0 w16a1e0b1 w12a0plain w12a1plain w16a0plain w16a1plain; do args="$args $n=$V/libtac_$n.so"; done
TAC_AB_N=200 python tools/r04/ab_inproc.py stft $args 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch28_ab_stft_sweep.txt
args=""
for n in w16a0e0b0 w12a0e1b0 w12a0e0b0 w12a0e0b1 w12a1e0b0 w12a1e0b1 w16a0e0b1 w16a1e0b0 w16a1e0b1 w12a0plain w12a1plain w16a0plain w16a1plain; do args="$args $n=$V/libtac_$n.so"; done
TAC_AB_N=200 python tools/r04/ab_inproc.py spec $args 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch28_ab_spec_sweep.txt
grep -v "^check" gpurun_out/r04/batch28_ab_stft_sweep.txt; grep -v "^check" gpurun_out/r04/batch28_ab_spec_sweep.txt
