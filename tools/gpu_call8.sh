cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests -m gpu -x -q -k "mel or golden or g0" 2>&1 | tail -3
for rep in 1 2 3; do
for rot in 1 4; do
  export TAC_ROTATE=$rot
  echo "== rotate $rot"
  TAC_S3_WAVES=12 python tools/time_steady.py mel 2>&1 | grep median | sed 's/^/s3x12  /'
  python tools/time_steady.py mel 2>&1 | grep median | sed 's/^/s3x15  /'
done
done | tee gpurun_out/ab/mel_s3_waves.txt
