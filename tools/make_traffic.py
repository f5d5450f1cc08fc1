#!/usr/bin/env python
"""profiles/<tag>_rocprofv3_summary.txt -> profiles/traffic.json: HBM bytes per launch per kernel.
FETCH_SIZE / WRITE_SIZE are KiB from separate rocprofv3 --pmc passes; FETCH_SIZE is doubled (gfx950
reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md §HBM)."""
import json
import re
import sys

tag, workload = sys.argv[1], sys.argv[2]
txt = open('profiles/%s_rocprofv3_summary.txt' % tag).read()


def sect(name):
    m = re.search(r'== %s \(mean per dispatch\) ==\n(.*?)(\n==|\Z)' % name, txt, re.S)
    return m.group(1)


def parse(block, key):
    out = {}
    for line in block.strip().splitlines():
        m = re.match(r"(.+?) \{.*'%s': ([0-9.]+)" % key, line)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


f, w = parse(sect('pmc_fetch'), 'FETCH_SIZE'), parse(sect('pmc_write'), 'WRITE_SIZE')
tr = {}
for k in f:
    name = 'k_scan' if 'k_scan' in k else 'k_stft' if 'k_stft' in k else 'k_pair' if ('k_pairmerge' in k or 'k_pairlane' in k) else k
    tr[name] = round((2 * f[k] + w.get(k, 0)) * 1024.0)
try:
    allj = json.load(open('profiles/traffic.json'))
except Exception:
    allj = {}
allj['_note'] = ('HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/<tag>_rocprofv3_summary.txt '
                 '(separate --pmc passes); FETCH_SIZE doubled per the gfx950 correction')
allj[workload] = tr
allj[workload + '_source'] = 'profiles/%s_rocprofv3_summary.txt' % tag
json.dump(allj, open('profiles/traffic.json', 'w'), indent=1)
print(tr)
