#!/usr/bin/env python
"""Timeline of bench.py's c4_job from a `rocprofv3 --kernel-trace --memory-copy-trace` run: the ten PCM uploads of the timed job
(DMA copies and the runtime's shader copies `__amd_rocclr_copyBuffer`), the spectral / scan kernels between them, link idle time.
Usage: python tools/c4_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import sys

d = sys.argv[1]
ks = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)[0])))
ev = []
for r in mc:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'HOST_TO_DEVICE' in r['Direction'] and e - s > 3e6:
        ev.append((s, e, 'upload (DMA engine)'))
for r in ks:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name']
    if 'copyBuffer' in n and e - s > 3e6:
        ev.append((s, e, 'upload (SHADER copy __amd_rocclr_copyBuffer)'))
    elif 'k_stft<short, true' in n or ('k_scan_small' in n and e - s > 2e5):
        ev.append((s, e, n.split('(')[0].replace('void ', '')))
ev.sort()
ups = [x for x in ev if x[2].startswith('upload')]
# the timed job = the ten big uploads before the last three (bench.py times three uploads on their own afterwards)
job = ups[-13:-3]
t0, t1 = job[0][0], job[-1][1]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in job:                        # union of the upload intervals
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('timed job: 10 uploads, first start .. last end %.2f ms; link busy %.2f ms, idle %.2f ms; shader copies %d of 10'
      % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, sum('SHADER' in x[2] for x in job)))
print('ms since the first upload started:')
for s, e, n in ev:
    if s < t0 or s > t1 + 3_000_000:
        continue
    print('%9.3f -> %9.3f  %7.3f ms  %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
