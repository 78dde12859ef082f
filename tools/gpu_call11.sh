cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for rot in 1 4; do
  export TAC_ROTATE=$rot
  unset TAC_AMD_LIB; python tools/time_steady.py stft spec 2>&1 | grep median | sed "s/^/nt  rot$rot /"
  export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_nont.so; python tools/time_steady.py stft spec 2>&1 | grep median | sed "s/^/plain rot$rot /"
done
done
