cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
TAC_AMD_LIB=$PWD/gpurun_variants/libtac_sttiming.so python tools/stream_timing.py > gpurun_out/c2/stream_timing.txt 2>&1
cat gpurun_out/c2/stream_timing.txt
