cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -q -m gpu -x -k "grad or autograd or fuzz or 400" 2>&1 | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c7/kt -o g -- python tools/prof_driver.py grad400h160 40 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c7/kt/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in list(csv.DictReader(open(f))):
    if 'tac::' in r['Name'] and int(r['Calls'])>=40:
        print(r['Name'][:60], r['Calls'], r['AverageNs']); tot+=float(r['TotalDurationNs'])/40e6
print('per step ms', tot)
PY
