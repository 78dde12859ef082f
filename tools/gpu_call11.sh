cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
TAC_STFT_S3_WAVES=12 TAC_STFT_S3_EARLY=1 timeout 600 python -m pytest tests -m gpu -x -q -k "stft or spectrogram or tiny or layout" 2>&1 | tail -2
for rep in 1 2; do
for rot in 1 4; do
  export TAC_ROTATE=$rot
  python tools/time_steady.py stft spec 2>&1 | grep median | sed "s/^/default(12c,16r) rot$rot /"
  TAC_STFT_S3_WAVES=12 python tools/time_steady.py stft spec 2>&1 | grep median | sed "s/^/12 late rot$rot /"
  TAC_STFT_S3_WAVES=12 TAC_STFT_S3_EARLY=1 python tools/time_steady.py stft spec 2>&1 | grep median | sed "s/^/12 early rot$rot /"
done
done | tee gpurun_out/ab/stft_early.txt
