cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in q4fly4 q4fly6 q4fly8; do
  export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so
  python tools/time_steady.py mel400 2>&1 | grep median | sed "s/^/s3 $v /"
done
done
