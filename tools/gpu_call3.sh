cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
python -m pytest tests -q -m gpu -x -k "mel or fused or lane or filterbank or fuzz or 400 or small" 2>&1 | tail -5 > gpurun_out/c3/pytest.log
bash tools/gpu_ab.sh "base default" "mel512 mel1024 mel400" > gpurun_out/c3/ab.txt 2>&1
for v in base default; do if [ $v = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi; python tools/time_others.py apply_filterbank 2>&1 | grep apply; done > gpurun_out/c3/fb.txt
cat gpurun_out/c3/pytest.log gpurun_out/c3/ab.txt gpurun_out/c3/fb.txt
