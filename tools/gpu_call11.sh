cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "mel or golden or coded or g9 or fuzz or cfg2 or cfg3" 2>&1 | tail -2
for rep in 1 2 3; do
  unset TAC_AMD_LIB; python tools/time_steady.py mel 2>&1 | grep median | sed "s/^/rotated /"
  export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_norot.so; python tools/time_steady.py mel 2>&1 | grep median | sed "s/^/in-order /"
done
