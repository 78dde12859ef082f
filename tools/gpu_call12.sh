cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/host
python tools/host_overhead.py 2>&1 | grep issue
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_host -o h -- python $GRAFT_REPO_ROOT/tools/host_overhead.py > /tmp/h.log 2>&1
python - <<'PY'
import csv,glob,statistics
f=glob.glob('/tmp/kt_host/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'melspec' in r['Kernel_Name']]
small=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if int(r['Grid_Size_X'])<=1024]
big=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if int(r['Grid_Size_X'])>1024]
print('small-input kernel: n=%d median %.1f us min %.1f us'%(len(small),statistics.median(small),min(small)))
print('cfg-2 kernel: n=%d median %.1f us'%(len(big),statistics.median(big)))
PY
