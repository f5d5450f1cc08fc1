#!/usr/bin/env python
"""GPU box: the chunked onset filter (k_hpf chunk mode) over many long files -- how often does its boundary check
fail (the unit then falls back to the sequential kernel), and are the rows ever different from the dense sequential path's?
Random clips of 95 .. 600 s: noise, tonal + gated, noise with stretches of digital silence, low-level noise, s16 and float32.

    python tools/hpf_chunk_stress.py [--clips 300] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audfprint_amd                                    # noqa: E402
audfprint_amd.configure_runtime()
from oracle import afp_oracle as O                      # noqa: E402  (synthetic clips only)
from audfprint_amd.batch import Extractor               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=300)
    ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    rng = np.random.RandomState(a.seed)
    ex = Extractor.get(0)
    ex.set_params()
    nfail = nbad = nchunked = frames = 0
    t0 = time.time()
    base = ex.extract(clips=[O.synth_noise(1, 1.0)]) and ex.path_stats()['hpf_chunked_total']
    for i in range(a.clips):
        secs = float(rng.uniform(95.0, 600.0))
        kind = rng.randint(5)
        if kind == 0:
            d = O.synth_noise(10000 + i, secs)
        elif kind == 1:
            d = O.synth_tonal(10000 + i, secs)
        elif kind == 2:
            d = O.synth_noise(10000 + i, secs).copy()
            for _ in range(rng.randint(1, 4)):
                s0 = rng.randint(0, len(d) - 11025)
                d[s0:s0 + rng.randint(2000, 30 * 11025)] = 0.0
        elif kind == 3:
            d = (O.synth_noise(10000 + i, secs) * np.float32(2.0 ** -rng.randint(1, 9))).astype(np.float32)
        else:
            d = np.round(O.synth_noise(10000 + i, secs) * 32768).astype(np.int16)
        ex.set_pipeline(compact=0, seg=1)
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        nfail += 1 if st['failed_units'] else 0
        ex.set_pipeline(compact=0, seg=0)
        q = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        if not (np.array_equal(r.hashes, q.hashes) and np.array_equal(r.peaks, q.peaks)):
            nbad += 1
            print('MISMATCH clip %d kind %d secs %.1f' % (i, kind, secs), flush=True)
        frames += 1 + len(d) // 256
    ex.set_pipeline()
    nchunked = ex.path_stats()['hpf_chunked_total'] - base
    print('%d clips (%d frames, %.0f s of audio), %d through the chunked filter: boundary check failed on %d (-> sequential kernel), '
          'rows differ from the dense path on %d; %.0f s' % (a.clips, frames, frames * 256 / 11025.0, nchunked, nfail, nbad, time.time() - t0))
    return 1 if nbad else 0


if __name__ == '__main__':
    sys.exit(main())
