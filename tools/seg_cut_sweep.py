#!/usr/bin/env python
"""GPU box: which (segment length, warm-up) should a file of a given length be cut with?  (VERDICT r4 #5b)
For clips of 10 .. 300 s (noise at three seeds, one gated tonal clip) and a list of cuts: ms per one-file call (hashes only,
100 / 20 calls), segments, segments the chain launch had to re-run (a warm-up that did not reach its neighbour's state), and
whether the rows equal the default cut's.  One JSON line per (clip, cut).   Usage: python tools/seg_cut_sweep.py [secs ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audfprint_amd                                      # noqa: E402
audfprint_amd.configure_runtime()
from audfprint_amd.batch import Extractor               # noqa: E402

SR = 11025
CUTS = ((0, 0), (64, 128), (64, 64), (48, 96), (48, 48), (40, 80), (32, 96), (32, 64), (24, 48), (96, 96), (128, 128))


def noise(seed, secs):
    rng = np.random.RandomState(seed)
    x = rng.randn(int(SR * secs)) * 0.1
    return (np.round(np.clip(x, -1, 1) * 32767).astype(np.int16).astype(np.float32) / np.float32(32768))


def tonal(seed, secs):
    """FM sinusoids under a 2 Hz square gate + a little noise, int16-quantised: digital-silence plateaus, loud/quiet stretches."""
    rng = np.random.RandomState(seed)
    n = int(SR * secs)
    t = np.arange(n) / SR
    x = np.zeros(n)
    for _ in range(12):
        f0, fd, fm = rng.uniform(200, 4500), rng.uniform(0, 80), rng.uniform(0.2, 3)
        x += rng.uniform(0.02, 0.1) * np.sin(2 * np.pi * f0 * t + fd / fm * np.sin(2 * np.pi * fm * t))
    x *= (np.sin(2 * np.pi * 2.0 * t) > 0)
    x += 0.001 * rng.randn(n) * (np.sin(2 * np.pi * 0.25 * t) > -0.5)
    return (np.round(np.clip(x, -1, 1) * 32767).astype(np.int16).astype(np.float32) / np.float32(32768))


def timed(fn, n):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    return (time.perf_counter() - t0) / n * 1e3, r


def main():
    secs_list = [float(a) for a in sys.argv[1:]] or [10.0, 20.0, 30.0, 60.0, 120.0, 300.0]
    ex = Extractor.get(0)
    ex.set_params()
    for secs in secs_list:
        clips = [('noise%d' % s, noise(s, secs)) for s in (77, 78, 79)] + [('tonal', tonal(5, secs))]
        n = 100 if secs <= 60 else 20
        for name, d in clips:
            ex.set_pipeline()
            _, r0 = timed(lambda: ex.extract(clips=[d], want_hashes=True, want_peaks=False), 1)
            for L, W in CUTS:
                if L:
                    ex.set_pipeline(seg=1, seg_len=L, seg_warm=W)
                else:
                    ex.set_pipeline()
                ms, rs = timed(lambda: ex.extract(clips=[d], want_hashes=True, want_peaks=False), n)
                st = ex.seg_stats()
                print(json.dumps(dict(secs=secs, clip=name, frames=1 + len(d) // 256, L=L, W=W, ms=round(ms, 4), used=st['used'],
                                      segments=st['segments'], rerun=st['rerun_fwd'] + st['rerun_bwd'], failed=st['failed_units'],
                                      same_rows=bool(np.array_equal(rs.hashes, r0.hashes)))), flush=True)
    ex.set_pipeline()


if __name__ == '__main__':
    main()
