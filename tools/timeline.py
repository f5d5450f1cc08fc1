"""Reduce a rocprofv3 kernel_trace.csv to a per-kernel timeline: duration of each dispatch, which other kernels
ran concurrently with it, and the steady-state period between successive k_stft launches.
Usage: python tools/timeline.py <dir-with-*kernel_trace.csv> [n_last_batches [skip_trailing_batches]]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.split('(')[0]
    for p in ('void ', ):
        if n.startswith(p):
            n = n[len(p):]
    return n.split('<')[0]


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
                             r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
    rows.sort()
    if not rows:
        print('no kernel trace rows under', d)
        return
    t0 = rows[0][0]
    stft = [r for r in rows if r[2] == 'k_stft']
    nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # ignore this many trailing k_stft launches
    if skip:
        cut = stft[-skip][0]
        rows = [r for r in rows if r[0] < cut]
        stft = stft[:-skip]
    # the compact pipeline launches k_stft twice per batch (the short second launch re-transforms the chunks of the units
    # that needed the floor): batches are counted by the long launches
    stft = [r for r in stft if r[1] - r[0] > 200000] or stft
    # bench.py runs the timed pipelined steps first and the same steps strictly one after another afterwards: the
    # steady-state window is taken from the PIPELINED part -- the k_stft launches that overlap a k_scan launch
    scans = [r for r in rows if r[2].startswith('k_scan')]
    def overlapped(r):
        return any(s[0] < r[1] and s[1] > r[0] for s in scans)
    ov = [r for r in stft if overlapped(r)]
    if len(ov) >= 4:
        last = ov[-1]
        rows = [r for r in rows if r[0] <= last[1]]
        stft = [r for r in stft if r[0] <= last[0]]
    # steady-state window: the last nlast k_stft launches (of the pipelined part when there is one)
    win0 = stft[-nlast][0] if len(stft) >= nlast else stft[0][0]
    sel = [r for r in rows if r[0] >= win0]
    print('dispatches in window: %d, window %.3f ms, k_stft launches %d' % (len(sel), (sel[-1][1] - win0) / 1e6, nlast))
    starts = [r[0] for r in stft[-nlast:]]
    if len(starts) > 1:
        per = [(b - a) / 1e6 for a, b in zip(starts[:-1], starts[1:])]
        print('k_stft start-to-start (ms):', ' '.join('%.3f' % p for p in per))
    agg = defaultdict(list)
    for s, e, n, q, st in sel:
        agg[n].append((e - s) / 1e6)
    print('%-24s %5s %9s %9s %9s' % ('kernel', 'n', 'mean ms', 'min', 'max'))
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print('%-24s %5d %9.4f %9.4f %9.4f' % (n, len(v), sum(v) / len(v), min(v), max(v)))
    # overlap matrix for the big kernels
    big = [r for r in sel if (r[1] - r[0]) > 100000]
    print('\ntimeline of kernels > 0.1 ms (ms since window start; q = queue id):')
    for s, e, n, q, st in big[:60]:
        conc = [b[2] for b in big if b is not None and b[0] < e and b[1] > s and (b[0], b[1], b[2]) != (s, e, n)]
        print('  %8.3f -> %8.3f  %-14s q%-3s  with: %s' % ((s - win0) / 1e6, (e - win0) / 1e6, n, q, ','.join(conc)))


if __name__ == '__main__':
    main()
