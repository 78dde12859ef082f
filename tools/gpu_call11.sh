cd $GRAFT_REPO_ROOT
export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_ldsx.so
timeout 600 python -m pytest tests -m gpu -x -q -k "g2 or golden or cfg2" 2>&1 | tail -2
for rep in 1 2 3; do
  unset TAC_AMD_LIB; python tools/time_steady.py mel 2>&1 | grep median | sed "s/^/permlane /"
  export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_ldsx.so; python tools/time_steady.py mel 2>&1 | grep median | sed "s/^/lds      /"
done
