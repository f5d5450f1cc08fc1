#!/usr/bin/env python
"""One-off, where a reference tree is at hand (AFP_REF_DIR): the REFERENCE's own find_peaks -> peaks2landmarks -> landmarks2hashes ->
unique/sort timed on the bench host's CPU next to the numpy oracle (bench.py's cpu_baseline, kind "port"), same clips, one
thread.  Needs the reference tree (AFP_REF_DIR)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ref = os.environ.get('AFP_REF_DIR', '/root/reference')
sys.path.insert(0, ref)
import audfprint_analyze as RA            # the reference module
from oracle import afp_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
clips = [O.synth_noise(1000003 * 0 + i, 30.0) for i in range(n)]
a = RA.Analyzer()
a.n_fft, a.n_hop, a.shifts = 512, 256, 1


def ref_hashes(d):
    pk = a.find_peaks(d, 11025)
    lm = a.peaks2landmarks(pk)
    h = RA.landmarks2hashes(lm)
    k = np.sort(np.unique((h[:, 0].astype(np.uint64) << np.uint64(32)) + h[:, 1].astype(np.uint64)))
    return np.stack([(k >> np.uint64(32)).astype(np.int32), (k & np.uint64(0xffffffff)).astype(np.int32)], axis=1)


t0 = time.perf_counter(); hr = [ref_hashes(d) for d in clips]; tr = time.perf_counter() - t0
t0 = time.perf_counter(); ho = [O.extract(d, O.Params())[1] for d in clips]; to = time.perf_counter() - t0
ok = all(np.array_equal(x, y) for x, y in zip(hr, ho))
nh = sum(len(x) for x in hr)
print('host cpus %d; %d clips x 30 s, one thread' % (os.cpu_count(), n))
print('reference (dpwe/audfprint Analyzer):  %.2f s  = %.0f x real time, %.0f hashes/s' % (tr, n * 30.0 / tr, nh / tr))
print('oracle    (oracle/afp_oracle.py):     %.2f s  = %.0f x real time, %.0f hashes/s' % (to, n * 30.0 / to, nh / to))
print('rows identical: %s' % ok)
