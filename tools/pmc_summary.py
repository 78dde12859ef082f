#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs: per-kernel mean of every counter.  python tools/pmc_summary.py dir..."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0][-60:]
            if 'tac::' not in r['Kernel_Name']:
                continue
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-32s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
