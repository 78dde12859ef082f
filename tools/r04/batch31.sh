#!/bin/bash
# backward kernel: nontemporal loads of the frame's oldest hop (last use) on top of the nontemporal stores / mel-gradient loads: time and L2-miss traffic
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V=gpurun_variants
python tools/r04/ab_inproc.py bwd shipped=torchaudio-contrib_amd/libtac_amd.so nt7=$V/libtac_br_nt7.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/batch31_ab_bwd_nt7.txt
cat gpurun_out/r04/batch31_ab_bwd_nt7.txt
for lib in torchaudio-contrib_amd/libtac_amd.so $V/libtac_br_nt7.so; do
  n=$(basename $lib .so)
  rm -rf /tmp/pf /tmp/pw
  TAC_AMD_LIB=$PWD/$lib rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o p -- python tools/prof_driver.py grad 3 > /dev/null 2>&1
  TAC_AMD_LIB=$PWD/$lib rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o p -- python tools/prof_driver.py grad 3 > /dev/null 2>&1
  python - <<PY
import csv,glob
for d,c in (('/tmp/pf','FETCH_SIZE'),('/tmp/pw','WRITE_SIZE')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        vals=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'backward_ring3' in r['Kernel_Name'] and r['Counter_Name']==c]
        if vals: print('$n', c, 'per launch: %.1f MB' % (sum(vals)/len(vals)*(2048 if c=='FETCH_SIZE' else 1024)/1e6), len(vals))
PY
done 2>&1 | tee gpurun_out/r04/batch31_traffic.txt
