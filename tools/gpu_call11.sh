cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in default hpss3 hpss4; do
  if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
  python tools/time_others.py "hpss k=31 (frame" "hpss k=17" "hpss k=9" 2>&1 | grep "TB/s" | sed "s/^/$v  /"
done
done | tee gpurun_out/ab/hpss_occ.txt
