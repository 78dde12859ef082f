#!/bin/bash
# round 4, GPU batch 2: RCCL tests, smoke, the new bench line, and the row-store pacing A/B of stft_stream3_kernel
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch2; mkdir -p $out
export TAC_ROTATE=4
timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -x -q -s 2>&1 | tail -25 > $out/pytest_rccl.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
timeout 900 python bench.py > $out/bench_N1.json 2> $out/bench_N1.err
for rep in 1 2; do
  for v in default vm0 r03; do
    for pol in nt plain; do
      if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
      export TAC_S3_STORES=$pol
      echo "lib=$v stores=$pol"; timeout 120 python tools/time_steady.py stft spec 2>&1 | grep median
    done
  done
done > $out/ab_stft.txt
cat $out/pytest_rccl.txt $out/smoke.txt; tail -c 1500 $out/bench_N1.err; head -c 6000 $out/bench_N1.json; echo; cat $out/ab_stft.txt
