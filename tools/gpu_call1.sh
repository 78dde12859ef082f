set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
python -m pytest tests -q -m gpu --maxfail=20 2>&1 | tail -40 > gpurun_out/c1/pytest.log
python tools/time_others.py > gpurun_out/c1/time_others.txt 2>&1
python tools/time_steady.py stft spec mel stft4096 spec4096 stft512 spec512 stft1024 spec1024 mel512 mel1024 mel400 stft400 spec400 mel256 > gpurun_out/c1/time_steady.txt 2>&1
python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
tail -5 gpurun_out/c1/pytest.log
