cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
python -m pytest tests -q -m gpu -x -k "400 or gradient or fuzz or autograd or announced" 2>&1 | tail -25 > gpurun_out/c4/pytest.log
cat gpurun_out/c4/pytest.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4/kt -o g400 -- python tools/prof_driver.py grad400 20 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c4/kt/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
