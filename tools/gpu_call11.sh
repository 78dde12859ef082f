cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in default q4fly4 q4fly8; do
  if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
  python tools/time_steady.py mel400 2>&1 | grep median | sed "s/^/$v /"
done
done
