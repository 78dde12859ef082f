#!/bin/bash
# round 4, GPU batch 10: host-side profile of one fused call; MFMA vs band-sparse contraction on today's kernels; round-4 profile collection
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch10; mkdir -p $out
timeout 200 python tools/host_profile.py > $out/host_profile.txt 2>&1
timeout 200 python tools/host_overhead.py > $out/host_overhead.txt 2>&1
export TAC_ROTATE=4
for rep in 1 2; do
  for path in auto mfma; do
    echo "TAC_MEL_PATH=$path"; TAC_MEL_PATH=$path timeout 200 python tools/time_steady.py mel mel1024 mel512 2>&1 | grep median
  done
done > $out/mfma_vs_sparse.txt
unset TAC_ROTATE
timeout 1500 bash tools/collect_profiles.sh r04 > $out/collect.log 2>&1
tail -40 $out/host_profile.txt; cat $out/host_overhead.txt $out/mfma_vs_sparse.txt; tail -5 $out/collect.log
