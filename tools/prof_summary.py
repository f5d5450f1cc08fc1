#!/usr/bin/env python
"""Summarise rocprofv3 output dirs written by tools/prof.sh into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, '**', pat), recursive=True))


try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from audfprint_amd import build as _b
    print('build_id: %s' % _b.source_id())          # the library these counters were taken from (bench.py checks it)
except Exception as e:       # noqa: BLE001
    print('build_id: unknown (%r)' % (e,))
print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
for f in find('stats', '*kernel_stats.csv'):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 14:
                print(','.join(row))
for sub in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write'):
    files = find(sub, '*counter_collection.csv')
    if not files:
        print('== %s: no counter csv ==' % sub)
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in files:
        with open(f) as fh:
            rd = csv.DictReader(fh)
            for row in rd:
                k = row.get('Kernel_Name', '?')
                k = k.split('(')[0]
                c = row.get('Counter_Name')
                v = float(row.get('Counter_Value', 0) or 0)
                agg[k][c] += v
                cnt[(k, c)] += 1
    print('== %s (mean per dispatch) ==' % sub)
    for k in sorted(agg):
        if not k.startswith('k_') and 'k_' not in k:
            continue
        print(k, {c: round(agg[k][c] / max(1, cnt[(k, c)]), 1) for c in sorted(agg[k])})
