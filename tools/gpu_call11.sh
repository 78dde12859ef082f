cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests -m gpu -x -q -k "mel or golden or g9 or coded" 2>&1 | tail -2
for rep in 1 2 3; do
for v in default late; do
  if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
  TAC_ROTATE=1 python tools/time_steady.py mel 2>&1 | grep median | sed "s/^/$v rot1 /"
  TAC_ROTATE=4 python tools/time_steady.py mel 2>&1 | grep median | sed "s/^/$v rot4 /"
done
done | tee gpurun_out/ab/early_first.txt
