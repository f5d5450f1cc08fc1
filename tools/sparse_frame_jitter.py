#!/usr/bin/env python
"""Which sparse frames are decided by FFT rounding noise IN THE REFERENCE ITSELF?  

Runs in the build container only (imports the live reference from /root/reference).  Method (the judge's): wrap np.fft.rfft
so that every bin is multiplied by 1 + 1e-15 * randn -- a perturbation two orders below the distance between numpy's
log-spectrogram and any other correct implementation's -- and count the peaks of Analyzer.find_peaks
(audfprint_analyze.py:255-308) that change.  A signal class whose peaks move under that jitter cannot be reproduced
bit-for-bit by anything but numpy's own pocketfft build; a class whose peaks never move can.

Signal: 1 s of digital silence holding the sparse samples from sample 5000 on, then 4 s of noise at -50 dB (the level at
which the click frames keep reference peaks through the backward pass)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, '/root/reference')
import audfprint_analyze as REF      # noqa: E402

_rfft = np.fft.rfft


def peaks_of(d, jitter_seed=None, density=20.0):
    an = REF.Analyzer(density)
    if jitter_seed is None:
        np.fft.rfft = _rfft
    else:
        rng = np.random.RandomState(jitter_seed)
        def jr(x, *a, **k):
            y = _rfft(x, *a, **k)
            return y * (1.0 + 1e-15 * rng.randn(*y.shape))
        np.fft.rfft = jr
    try:
        p = an.find_peaks(d, 11025)
    finally:
        np.fft.rfft = _rfft
    return set((int(t), int(b)) for t, b in p)


def signal(pos_amp, tail_db=-50.0, seed=1234):
    rng = np.random.RandomState(seed)
    x = np.concatenate([np.zeros(11025), rng.randn(4 * 11025) * 10 ** (tail_db / 20.0)])
    for p, a in pos_amp:
        x[p] = a
    return (np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)).astype(np.float32) / np.float32(32768)


def trial(pos_amp, tail_db=-50.0):
    d = signal(pos_amp, tail_db)
    base = peaks_of(d)
    return len(base), [len(base ^ peaks_of(d, s)) for s in (1, 2, 3)]


def main():
    out = []
    def rec(label, pos_amp, tail_db=-50.0):
        n, diffs = trial(pos_amp, tail_db)
        par = sorted(set(p & 1 for p, _ in pos_amp))
        r = dict(case=label, n_nonzero=len(pos_amp), parities=par, tail_db=tail_db, ref_peaks=n, differing_under_jitter=diffs)
        out.append(r)
        print(json.dumps(r), flush=True)
    rec('one', [(5000, 0.5)])
    for dlt in (1, 3, 63, 64, 100, 101, 127, 128, 255, 256, 257, 300, 301):
        rec('two equal, delta %d' % dlt, [(5000, 0.5), (5000 + dlt, 0.5)])
        rec('two 0.5/0.25, delta %d' % dlt, [(5000, 0.5), (5000 + dlt, 0.25)])
    rec('three equal, spacing 128', [(5000 + 128 * i, 0.5) for i in range(3)])
    rec('four equal, spacing 64', [(5000 + 64 * i, 0.5) for i in range(4)])
    rec('four equal, spacing 63', [(5000 + 63 * i, 0.5) for i in range(4)])
    rng = np.random.RandomState(7)
    for K in (3, 4, 6, 8, 16, 32):
        for tr in range(6):
            pos = 5000 + 2 * np.sort(rng.choice(128, K, replace=False))          # all even offsets
            amp = rng.uniform(0.05, 0.9, K) * rng.choice([-1, 1], K)
            rec('K=%d same parity #%d' % (K, tr), list(zip(pos.tolist(), amp.tolist())))
        for tr in range(6):
            while True:
                pos = 5000 + np.sort(rng.choice(256, K, replace=False))
                if len(set((pos & 1).tolist())) == 2:
                    break
            amp = rng.uniform(0.05, 0.9, K) * rng.choice([-1, 1], K)
            rec('K=%d mixed parity #%d' % (K, tr), list(zip(pos.tolist(), amp.tolist())))
    for lbl, pa in (('one', [(5000, 0.5)]), ('two equal, delta 256', [(5000, 0.5), (5256, 0.5)]), ('two equal, delta 100', [(5000, 0.5), (5100, 0.5)])):
        rec(lbl, pa, tail_db=-20.0)
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'r05_sparse_frame_jitter_reference.json')
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
