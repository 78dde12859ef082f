#!/bin/bash
# round 4, GPU batch 11: blocks per CU of the elementwise kernels (write-heavy ones may prefer fewer, like a pure fill)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch11; mkdir -p $out
for rep in 1 2; do
for n in 8 2 3 4 16; do
  echo "TAC_EW_BLOCKS_PER_CU=$n"; TAC_EW_BLOCKS_PER_CU=$n timeout 200 python tools/time_others.py complex_norm magphase amplitude_to_db db_to_amplitude mu_law 2>&1 | grep -v amdgpu.ids
done
done > $out/ew_blocks.txt
cat $out/ew_blocks.txt
