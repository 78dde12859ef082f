cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in default fbl12 fbl16 fbl16b; do
  if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
  python tools/time_others.py apply_filterbank 2>&1 | grep "TB/s" | sed "s/^/$v  /"
  python tools/time_steady.py mel4096 2>&1 | grep median | sed "s/^/$v  /"
done
done | tee gpurun_out/ab/fbl.txt
