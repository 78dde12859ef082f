cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
  TAC_SMALL2=1 python tools/time_steady.py stft512 spec512 mel512 stft1024 spec1024 mel1024 2>&1 | grep median | sed 's/^/two-wave  /'
  TAC_SM3_WAVES=12 python tools/time_steady.py stft512 spec512 mel512 stft1024 spec1024 mel1024 2>&1 | grep median | sed 's/^/s3x12     /'
  TAC_SM3_WAVES=16 python tools/time_steady.py stft512 spec512 stft1024 spec1024 2>&1 | grep median | sed 's/^/s3x16     /'
done | tee gpurun_out/ab/small3.txt
