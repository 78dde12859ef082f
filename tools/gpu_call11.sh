cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "4096 or cfg4 or tiny or fuzz or golden or g4 or layout" 2>&1 | tail -3
for rep in 1 2 3; do
  python tools/time_steady.py stft4096 spec4096 mel4096 2>&1 | grep median
done
