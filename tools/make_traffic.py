#!/usr/bin/env python
"""profiles/<tag>_rocprofv3_summary.txt -> profiles/traffic.json (HBM bytes per launch per kernel) and profiles/pmc.json
(VALU busy quad-cycles per step), which bench.py reads for roofline.traffic / roofline.valu_issue.
FETCH_SIZE / WRITE_SIZE are KiB from separate rocprofv3 --pmc passes; FETCH_SIZE is doubled (gfx950 reports half the bytes
of wide coalesced reads, MI355X_MICROARCH.md §HBM).   Usage: python tools/make_traffic.py <tag> <workload key>"""
import json
import re
import sys

tag, workload = sys.argv[1], sys.argv[2]
txt = open('profiles/%s_rocprofv3_summary.txt' % tag).read()


def sect(name):
    m = re.search(r'== %s \(mean per dispatch\) ==\n(.*?)(\n==|\Z)' % name, txt, re.S)
    return m.group(1) if m else ''


def parse(block, key):
    out = {}
    for line in block.strip().splitlines():
        m = re.match(r"(.+?) \{.*'%s': ([0-9.]+)" % key, line)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


def short(k):
    return ('k_scan' if 'k_scan' in k else 'k_stft' if 'k_stft' in k else
            'k_pair' if ('k_pairmerge' in k or 'k_pairlane' in k) else k)


m_id = re.search(r'^build_id: (\S+)', txt, re.M)
build_id = m_id.group(1) if m_id else None

f, w = parse(sect('pmc_fetch'), 'FETCH_SIZE'), parse(sect('pmc_write'), 'WRITE_SIZE')
tr = {}
for k in f:
    if 'k_clock_probe' in k:
        continue
    # several kernels share a short name (compact + dense STFT, compact scan): one step launches each of them once
    tr[short(k)] = tr.get(short(k), 0) + round((2 * f[k] + w.get(k, 0)) * 1024.0)


def load(path):
    try:
        return json.load(open(path))
    except Exception:
        return {}


allj = load('profiles/traffic.json')
allj['_note'] = ('HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/<tag>_rocprofv3_summary.txt '
                 '(separate --pmc passes); FETCH_SIZE doubled per the gfx950 correction')
allj[workload] = tr
allj[workload + '_source'] = 'profiles/%s_rocprofv3_summary.txt' % tag
allj[workload + '_build_id'] = build_id
json.dump(allj, open('profiles/traffic.json', 'w'), indent=1)

va = parse(sect('pmc_sq'), 'SQ_ACTIVE_INST_VALU')
vi = parse(sect('pmc_sq'), 'SQ_INSTS_VALU')
sa = parse(sect('pmc_sq2'), 'SQ_INSTS_SALU')
pm = load('profiles/pmc.json')
pm['_note'] = ('per step (one launch of every kernel): SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles) of VALU '
               'busy time summed over the SIMDs; bench.py divides by 1024 SIMDs x step time x measured shader clock')
keep = {k: v for k, v in va.items() if 'k_clock_probe' not in k}
per_k = {}
for k, v in keep.items():
    per_k[short(k)] = per_k.get(short(k), 0.0) + v
pm[workload] = dict(valu_quad_cycles=sum(keep.values()), valu_insts=sum(v for k, v in vi.items() if 'k_clock_probe' not in k),
                    salu_insts=sum(v for k, v in sa.items() if 'k_clock_probe' not in k),
                    per_kernel_valu_quad_cycles=per_k, build_id=build_id,
                    source='profiles/%s_rocprofv3_summary.txt' % tag)
json.dump(pm, open('profiles/pmc.json', 'w'), indent=1)
print(tr)
print(pm[workload])
