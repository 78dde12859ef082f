cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in default fly8 fly10 fly12 fly16; do
  if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
  python tools/time_steady.py mel512 mel1024 2>&1 | grep median | sed "s/^/$v /"
done
done | tee gpurun_out/ab/sm_fly.txt
