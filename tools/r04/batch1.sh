#!/bin/bash
# round 4, GPU batch 1: correctness of the lean set-up + same-box A/B of the LDS knobs of melspec_stream3_kernel and the row-store policy
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch1; mkdir -p $out
export TAC_ROTATE=4
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $out/pytest_default.txt
for v in b64 nofence addtid w0 ptw all3; do
  TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "g1 or g2 or g9 or cfg2 or mel" 2>&1 | tail -2 > $out/pytest_$v.txt
done
for rep in 1 2 3; do
  for v in default r03 b64 nofence addtid w0 ptw all3; do
    if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
    timeout 120 python tools/time_steady.py mel 2>&1 | grep median
  done
done > $out/ab_mel.txt
unset TAC_AMD_LIB
for rep in 1 2; do
  for pol in auto nt plain; do
    if [ "$pol" = auto ]; then unset TAC_S3_STORES; else export TAC_S3_STORES=$pol; fi
    echo "stores=$pol"; timeout 120 python tools/time_steady.py stft spec 2>&1 | grep median
  done
  unset TAC_S3_STORES
  echo "r03 lib"; TAC_AMD_LIB=$PWD/gpurun_variants/libtac_r03.so timeout 120 python tools/time_steady.py stft spec 2>&1 | grep median
done > $out/ab_stft.txt
cat $out/pytest_*.txt; cat $out/ab_mel.txt $out/ab_stft.txt
