#!/usr/bin/env python
"""Developer diagnostic (GPU box): compare every stage of the HIP path with the oracle and
print where they first differ.  Usage: python tools/stage_check.py [golden-name ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, golden_names  # noqa: E402
from oracle import afp_oracle as O  # noqa: E402
from audfprint_amd.batch import Extractor  # noqa: E402


def check_case(ex, name, verbose=True):
    g = load_golden(name)
    p = g['params']
    prm = O.Params(**{k: p[k] for k in ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts',
                                        'targetdf', 'mindt', 'targetdt')})
    ex.set_params(**{k: p[k] for k in ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts',
                                       'targetdf', 'mindt', 'targetdt')})
    d = g['d']
    t0 = time.time()
    r = ex.extract(clips=[d], want_hashes=True, want_peaks=True, debug=True)
    dt = time.time() - t0
    ok = True
    msgs = []
    S = ex.shifts
    for s in range(S):
        pk = r.unit_peaks(0, s)
        if not np.array_equal(pk, g['peaks'][s]):
            ok = False
            msgs.append('shift %d peaks differ: got %d want %d' % (s, len(pk), len(g['peaks'][s])))
    if not np.array_equal(r.clip_hashes(0), g['hashes']):
        ok = False
        msgs.append('hashes differ: got %d want %d' % (len(r.clip_hashes(0)), len(g['hashes'])))
    line = '%-32s %s  peaks=%s hashes=%d  (%.1f ms)' % (name, 'OK ' if ok else 'BAD', [len(r.unit_peaks(0, s)) for s in range(S)],
                                                         len(r.clip_hashes(0)), dt * 1e3)
    print(line)
    if ok and not verbose:
        return ok
    # stage comparison on shift 0
    offs = O.shift_offsets(prm.shifts)
    st = O.find_peaks_stages(d[offs[0]:], prm)
    if 'mag' not in st:
        return ok
    T = st['mag'].shape[1]
    logS = ex.debug(0, np.float64, (256,))[:T]
    nyq = ex.debug(1, np.float64)[:T]
    stats = ex.debug(4, np.float64, (4,))
    with np.errstate(divide='ignore'):
        ref_log = np.log(st['mag'])
    fin = np.isfinite(ref_log[:256].T)
    e_log = np.max(np.abs(logS[fin] - ref_log[:256].T[fin])) if fin.any() else 0.0
    finn = np.isfinite(ref_log[256])
    e_nyq = np.max(np.abs(nyq[finn] - ref_log[256][finn])) if finn.any() else 0.0
    print('   T=%d  max|logS err|=%.3e  nyq err=%.3e  inf-mismatch=%d' % (T, e_log, e_nyq, int(np.sum(np.isfinite(logS) != fin))))
    if not st['zero']:
        mag = st['mag']
        ref_lf = np.log(mag.max() / 1e6)
        ref_mean = np.mean(np.log(np.maximum(mag, mag.max() / 1e6)))
        print('   logfloor dev=%.17g ref=%.17g | mean dev=%.17g ref=%.17g (diff %.3e)' % (stats[0, 0], ref_lf, stats[0, 1], ref_mean, stats[0, 1] - ref_mean))
    sg = ex.debug(2, np.float64, (256,))[:T]
    e_sg = np.max(np.abs(sg - st['sgram'].T))
    print('   max|HPF sgram err|=%.3e' % e_sg)
    cand = ex.debug(3, np.int32, (ex.K,))[:T]
    fwd = st['fwd']
    bad = 0
    first = None
    for t in range(T):
        a = set(int(b) for b in cand[t] if b >= 0)
        b = set(int(x) for x in np.nonzero(fwd[:, t])[0])
        if a != b:
            bad += 1
            if first is None:
                first = (t, sorted(a), sorted(b))
    print('   fwd candidate frames differing: %d  first=%s' % (bad, first))
    for m in msgs:
        print('   ' + m)
    if not ok:
        pk = r.unit_peaks(0, 0)
        want = g['peaks'][0]
        n = min(len(pk), len(want))
        dif = np.nonzero(np.any(pk[:n] != want[:n], axis=1))[0]
        if len(dif):
            i = dif[0]
            print('   first peak diff at %d: got %s want %s' % (i, pk[max(0, i - 1):i + 3].tolist(), want[max(0, i - 1):i + 3].tolist()))
    return ok


def main():
    names = sys.argv[1:] or golden_names()
    ex = Extractor.get(0)
    nbad = 0
    for nm in names:
        try:
            if not check_case(ex, nm, verbose=True):
                nbad += 1
        except Exception as e:  # keep going: this is a diagnostic
            nbad += 1
            print('%-32s EXC %r' % (nm, e))
    print('cases=%d bad=%d' % (len(names), nbad))
    return 1 if nbad else 0


if __name__ == '__main__':
    sys.exit(main())
