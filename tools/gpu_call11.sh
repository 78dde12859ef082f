cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "mel or golden or coded or g9 or fuzz or cfg2" 2>&1 | tail -2
TAC_AMD_LIB=$PWD/gpurun_variants/libtac_s3cyc.so python tools/stream3_cycles.py 2>&1 | tail -2
for rep in 1 2 3; do python tools/time_steady.py mel 2>&1 | grep median; done
