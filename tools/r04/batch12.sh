#!/bin/bash
# round 4, GPU batch 12: backward kernel without its loads / without its stores (timing-only ablations); elementwise block counts re-check
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch12; mkdir -p $out
V=$PWD/gpurun_variants
timeout 300 python tools/r04/ab_inproc.py bwd base=$PWD/torchaudio-contrib_amd/libtac_amd.so noload=$V/libtac_bwd_noload.so nostore=$V/libtac_bwd_nostore.so 2>&1 | grep -v amdgpu.ids > $out/ab_bwd_abl.txt
timeout 200 python tools/time_others.py complex_norm magphase mu_law 2>&1 | grep -v amdgpu.ids > $out/time_others_ew.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mulaw or complex_norm or magphase or elementwise" 2>&1 | tail -2 > $out/pytest_ew.txt
cat $out/ab_bwd_abl.txt $out/time_others_ew.txt $out/pytest_ew.txt
