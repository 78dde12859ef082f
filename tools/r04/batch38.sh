#!/bin/bash
# hpss: XCD-aware tile mapping (every XCD gets a contiguous eighth of the tiles) vs tile = block; also on the persistent form
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "hpss or g8" 2>&1 | tail -3 > gpurun_out/r04/batch38_tests.txt
TAC_FUZZ_CASES=300 TAC_FUZZ_SEED=11 python -m pytest tests/test_gpu_fuzz.py -x -q -k hpss 2>&1 | tail -3 >> gpurun_out/r04/batch38_tests.txt
V=gpurun_variants
for k in 31 9 5x9; do
python tools/r04/ab_other.py hpss:$k block=$V/libtac_hp_noxcd.so xcd=$V/libtac_hp_xcd.so xcd_persist=$V/libtac_hp_xcd_p3.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/batch38_ab_hpss_xcd.txt
cat gpurun_out/r04/batch38_tests.txt gpurun_out/r04/batch38_ab_hpss_xcd.txt
for lib in $V/libtac_hp_noxcd.so $V/libtac_hp_xcd.so; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  TAC_AMD_LIB=$PWD/$lib rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o p -- python tools/time_others.py "hpss k=31 (frame" > /dev/null 2>&1
done
python - <<PY
import csv,glob
for c,m in (('FETCH_SIZE',2048),('WRITE_SIZE',1024)):
    for f in glob.glob('/tmp/pm_%s/**/*counter_collection.csv' % c, recursive=True):
        v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name']==c and 'hpss' in r['Kernel_Name']]
        if v: print('$lib', c, 'per launch: %.1f MB over %d launches' % (sum(v)/len(v)*m/1e6, len(v)))
PY
done | tee gpurun_out/r04/batch38_hpss_traffic.txt
