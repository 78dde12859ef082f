#!/bin/bash
# hpss: HBM traffic per launch from the counters (separate passes), shipped kernel
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o p -- python tools/time_others.py "hpss k=31 (frame" > /dev/null 2>&1
done
python - <<'PY' | tee gpurun_out/r04/batch37_hpss_traffic.txt
import csv,glob
for c,m in (('FETCH_SIZE',2048),('WRITE_SIZE',1024)):
    for f in glob.glob('/tmp/pm_%s/**/*counter_collection.csv' % c, recursive=True):
        vals={}
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']==c and 'hpss' in r['Kernel_Name']:
                vals.setdefault(r['Kernel_Name'][:60],[]).append(float(r['Counter_Value']))
        for k,v in vals.items(): print(c, k, 'per launch: %.1f MB over %d launches' % (sum(v)/len(v)*m/1e6, len(v)))
PY
