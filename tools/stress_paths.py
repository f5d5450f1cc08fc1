"""Stress run (GPU): random ragged batches through the COMPACT and the SEGMENT-parallel paths against the dense path of the
same library, bit for bit -- looking for rare failures of the chunk-to-chunk filter hand-off (k_stft<ST,true>: flag / state
through HBM between workgroups that run concurrently) and of the segment boundaries.  Not a parity test (dense is checked
against the oracle by tests/): a repeatability test over many launch shapes.

    python tools/stress_paths.py [--iters 200] [--seed 1] [--secs 150]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--secs', type=float, default=150.0, help='stop after this many seconds')
    a = ap.parse_args()
    from audfprint_amd.batch import Extractor
    ex = Extractor.get(0)
    rng = np.random.default_rng(a.seed)
    pool = rng.standard_normal(11025 * 400).astype(np.float32) * 0.1
    # a tonal stretch and a silent stretch inside the pool: ties, floors and empty frames
    t = np.arange(11025 * 20) / 11025.0
    pool[11025 * 100:11025 * 120] = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.25 * np.sin(2 * np.pi * 1320 * t)).astype(np.float32)
    pool[11025 * 200:11025 * 205] = 0.0
    t0 = time.time()
    bad = 0
    n_units = 0
    it = 0
    seg_runs = seg_fallback_units = seg_reruns = seg_total = 0
    by_type = {}
    for it in range(a.iters):
        if time.time() - t0 > a.secs:
            break
        kind = it % 4
        if kind == 0:                       # many short clips (the chunk list is long, every unit 1-3 chunks)
            n, lo, hi = int(rng.integers(800, 3000)), 0.3, 6.0
        elif kind == 1:                     # the bench shape with ragged lengths
            n, lo, hi = int(rng.integers(768, 1400)), 3.0, 30.0
        elif kind == 2:                     # few long clips: long hand-off chains
            n, lo, hi = int(rng.integers(770, 900)), 20.0, 60.0
        else:                               # tiny units mixed in
            n, lo, hi = int(rng.integers(1000, 2500)), 0.01, 12.0
        lens = (rng.uniform(lo, hi, n) * 11025).astype(np.int64) + 1
        offs = rng.integers(0, len(pool) - int(lens.max()) - 1, n)
        clips = [pool[o:o + l] for o, l in zip(offs, lens)]
        # the sample type both runs ingest (round 6: the s16 / float64 / list instantiations of k_stft were rewritten)
        ingest = ('float32', 'float32', 'int16', 'int16', 'float64')[it % 5]
        if ingest == 'int16':
            clips = [np.round(np.clip(c, -1, 1) * 32767).astype(np.int16) for c in clips]
        elif ingest == 'float64':
            clips = [c.astype(np.float64) for c in clips]
        by_type[ingest] = by_type.get(ingest, 0) + 1
        shifts = int(rng.choice([1, 1, 1, 2, 4]))
        dens = float(rng.choice([20.0, 20.0, 70.0]))
        ex.set_params(density=dens, shifts=shifts)
        ex.set_pipeline(compact=0, seg=0)
        r0 = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        h0, o0, p0, f0 = r0.hashes.copy(), r0.hash_offsets.copy(), r0.peaks.copy(), r0.unit_flags.copy()
        ex.set_pipeline(compact=1, seg=0)
        r1 = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        ok = (np.array_equal(h0, r1.hashes) and np.array_equal(o0, r1.hash_offsets) and np.array_equal(p0, r1.peaks)
              and np.array_equal(f0 & 0x1f, r1.unit_flags & 0x1f))
        n_units += n * shifts
        if not ok:
            bad += 1
            print('MISMATCH compact vs dense: iter', it, 'n', n, 'shifts', shifts, 'density', dens, ingest, flush=True)
        if kind == 2 and shifts == 1:
            sub = clips[:100]
            ex.set_pipeline(compact=0, seg=0)
            ra = ex.extract(clips=sub, want_hashes=True, want_peaks=True)
            ha, pa = ra.hashes.copy(), ra.peaks.copy()
            ex.set_pipeline(compact=0, seg=1)
            rb = ex.extract(clips=sub, want_hashes=True, want_peaks=True)
            st = ex.seg_stats()
            seg_runs += 1
            seg_fallback_units += st['failed_units']
            seg_reruns += st['rerun_fwd'] + st['rerun_bwd']
            seg_total += st['segments']
            if not (np.array_equal(ha, rb.hashes) and np.array_equal(pa, rb.peaks)):
                bad += 1
                print('MISMATCH segments vs dense: iter', it, st, flush=True)
    ex.set_pipeline()
    ex.set_params()
    print('stress: %d iterations, %d units, %d mismatches, %.0f s; batches by ingest type %s' % (it + 1, n_units, bad, time.time() - t0, by_type))
    print('segments: %d batches of 100 clips, %d segments, %d re-runs, %d units re-done by the sequential kernel'
          % (seg_runs, seg_total, seg_reruns, seg_fallback_units))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
