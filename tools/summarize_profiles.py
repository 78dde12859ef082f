#!/usr/bin/env python
"""gpurun_out/<tag>/ (from tools/collect_profiles.sh) -> profiles/<tag>/: kernel_stats.csv (rocprofv3
--kernel-trace --stats of the bench command), pmc_<kernel>.json (per-launch means of every counter, HBM
traffic corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE is in KB and reports 1/2 of the bytes of a
wide coalesced stream on gfx950 -> x2048; WRITE_SIZE in KB -> x1024), bench_N1.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
src = os.path.join('gpurun_out', tag)
dst = os.path.join('profiles', tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(src + '/kt/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, 'kernel_stats.csv'), 'w') as o:
        w = csv.writer(o)
        for r in rows:
            if r and (r[0] == 'Name' or 'tac::' in r[0] or len(r[0]) < 160):
                w.writerow([r[0][:160]] + r[1:])
            else:
                w.writerow([r[0][:100] + '...'] + r[1:])
for f in glob.glob(src + '/kt_stages/**/*kernel_stats.csv', recursive=True):
    rows = [r for r in csv.reader(open(f)) if r and (r[0] == 'Name' or 'tac::' in r[0])]
    with open(os.path.join(dst, 'kernel_stats_stages.csv'), 'w') as o:
        csv.writer(o).writerows([[r[0][:160]] + r[1:] for r in rows])
if os.path.exists(src + '/bench_N1.json'):
    shutil.copy(src + '/bench_N1.json', dst + '/bench_N1.json')
for sub, name in (('kt_grad', 'kernel_stats_backward.csv'), ('kt_gradf', 'kernel_stats_backward_fused_op.csv'),
                  ('kt_grad400', 'kernel_stats_backward_n400.csv'), ('kt_gradspec', 'kernel_stats_backward_spectrogram.csv'),
                  ('kt_grad1024', 'kernel_stats_backward_n1024.csv'), ('kt_grad512', 'kernel_stats_backward_n512.csv')):
    for f in glob.glob(src + '/' + sub + '/**/*kernel_stats.csv', recursive=True):
        rows = [r for r in csv.reader(open(f)) if r and (r[0] == 'Name' or 'tac::' in r[0])]
        with open(os.path.join(dst, name), 'w') as o:
            csv.writer(o).writerows([[r[0][:140]] + r[1:] for r in rows])
os.makedirs(os.path.join(dst, 'steady_state'), exist_ok=True)
for f in ('time_steady.txt', 'time_others.txt', 'mel4096_banks.txt'):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, 'steady_state', f))
for k in ('mel', 'stft', 'spec', 'spec4096', 'mel4096', 'fb', 'grad'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in [f for sub in ('fetch', 'write', 'sq', 'mfma', 'stall') for f in glob.glob(src + '/pmc_%s_%s/**/*counter_collection.csv' % (k, sub), recursive=True)]:
        for r in csv.DictReader(open(f)):
            if 'tac::' in r['Kernel_Name'] and 'plan' not in r['Kernel_Name']:
                agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {}
    for name, cs in agg.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        if 'FETCH_SIZE' in d:
            d['hbm_read_bytes_corrected'] = d['FETCH_SIZE'] * 2048.0
        if 'WRITE_SIZE' in d:
            d['hbm_write_bytes'] = d['WRITE_SIZE'] * 1024.0
        if 'hbm_read_bytes_corrected' in d and 'hbm_write_bytes' in d:
            d['hbm_traffic_bytes_per_launch'] = d['hbm_read_bytes_corrected'] + d['hbm_write_bytes']
        out[name] = d
    if out:
        json.dump(out, open(os.path.join(dst, 'pmc_%s.json' % ('backward' if k == 'grad' else k)), 'w'), indent=1, sort_keys=True)
        for name, d in out.items():
            print(k, name, 'traffic/launch: %.1f MB' % (d.get('hbm_traffic_bytes_per_launch', float('nan')) / 1e6))
