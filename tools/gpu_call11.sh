cd $GRAFT_REPO_ROOT
for env in "TAC_STREAM2=1" "TAC_S3_WAVES=15" "TAC_STFT_PIPE2=1" "TAC_SMALL2=1" "TAC_BWD_LDS_RING=1" "TAC_STFT_S3_WAVES=12" "TAC_SM3_WAVES=12"; do
  echo "== $env"
  env $env timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
done
