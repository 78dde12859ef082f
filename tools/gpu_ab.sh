# usage: bash tools/gpu_ab.sh "<variant list>" "<time_steady names>"   (variants: names under gpurun_variants/libtac_<name>.so, or 'default')
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in $1; do
  if [ "$v" = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
  python tools/time_steady.py $2 2>&1 | grep median
done
done | tee gpurun_out/ab/last.txt
