#!/usr/bin/env python
"""VERDICT r5 #3 / SURVEY §7.2 step 9, sized on the CPU before any kernel is written: would a float32 SPECTRAL stage
(FFT, |S|^2, log -- half the FP64 instruction count, packed math) behind a parity gate pay?

The experiment (numpy / scipy only; the oracle is the float64 reference restatement):
  * spectral stage in float32: frames * window, scipy's single-precision rfft (complex64), |S|^2, log, floor -- all float32;
    mean and onset filter either in float64 over the float32 logs (`filter64`, what a kernel would do: 3 flops per value) or
    in float32 as well (`filter32`);
  * the two threshold passes in float64 over those values, with a GUARD: every decisive comparison (local maximum, `> sthresh`
    forward :217, the top-K cut :220-221, `>= sthresh` backward :242) is checked against a PER-VALUE error bound
        e(k,t) = u * (c1 * ||frame_t * window||_2 / |S(k,t)| + c2 * (1 + |log S|)),     u = 2^-24
    pushed through the filter (y[t] = x[t] - 0.02 * sum_j 0.98^(j-1) x[t-j]: worst-case gain 2) and carried by the thresholds
    (a bump inherits its source's bound, scaled by the Gaussian and the decay); c1, c2 are calibrated on the observed
    float32-vs-float64 differences with a safety factor;
  * a unit is FLAGGED if any such comparison falls inside its margin -- a flagged unit would be re-run in float64.
Reported per workload: units / frames flagged, units whose integers (peaks) differ from the float64 oracle, and -- the
soundness check -- whether every differing unit was flagged.  KILL CRITERION: more than 15 % of C3 units flagged.

    python tools/f32_screen_feasibility.py [--clips 256] [--out profiles/r06_f32_screen_feasibility.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.fft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import afp_oracle as O          # noqa: E402   (tools/ measurement helper: not product code)

U = 2.0 ** -24
N_FFT, N_HOP = 512, 256


def spectral_f32(d):
    """(|S| float32 (257,T), frame norms float64 (T,)): the float32 spectral stage up to the magnitude."""
    d32 = np.asarray(d, dtype=np.float32)
    x = np.pad(d32, N_FFT // 2, mode='reflect')
    nfr = 1 + (x.shape[0] - N_FFT) // N_HOP
    idx = (np.arange(nfr)[:, None] * N_HOP) + np.arange(N_FFT)[None, :]
    w32 = O.hann_window(N_FFT).astype(np.float32)
    fr = x[idx] * w32                                              # float32 * float32
    S = scipy.fft.rfft(fr, N_FFT, axis=1)                          # complex64: single-precision pocketfft
    assert S.dtype == np.complex64
    p = (S.real * S.real + S.imag * S.imag).astype(np.float32)     # |S|^2 in float32
    nrm = np.sqrt(np.sum(fr.astype(np.float64) ** 2, axis=1))
    return p.T, nrm


def logs_f32(p32):
    """float32 log-magnitude with the reference's floor (audfprint_analyze.py:283-285): 0.5 * log(max(p, pmax / 1e12))."""
    pmax = np.float32(p32.max())
    if not pmax > 0:
        return None
    fl = np.float32(pmax / np.float32(1e12))
    return (np.float32(0.5) * np.log(np.maximum(p32, fl))).astype(np.float32), fl


def value_bound(p32, nrm, logv, c1, c2):
    """e(k,t): bound on |log32 - log64| of every value (float64 array (257,T))."""
    mag = np.sqrt(np.maximum(p32.astype(np.float64), 1e-300))
    return U * (c1 * nrm[None, :] / mag + c2 * (1.0 + np.abs(logv.astype(np.float64))))


def hpf(x, dtype):
    """the onset filter of hpf_rows in `dtype` (float32: every operation rounded to single)"""
    pole = dtype(O.HPF_POLE)
    x = np.ascontiguousarray(x[:-1, :].astype(dtype))
    y = np.empty_like(x)
    z = np.zeros(x.shape[0], dtype)
    for n in range(x.shape[1]):
        xn = x[:, n]
        yn = xn + z
        z = xn * dtype(-1.0) - yn * (-pole)
        y[:, n] = yn
    return y


def filter_bound(e, emean, filt32, y):
    """E(k,t): bound on the filtered value's error.  Telescoped form: y[t] = x[t] - 0.02 * sum_{j>=1} 0.98^(j-1) x[t-j], so the
    input errors contribute e[t] + 0.02 * w[t], w[t] = e[t-1] + 0.98 w[t-1]; the mean's error (a constant shift of every x)
    contributes |dmean| * 0.98^t <= |dmean|; a float32 recurrence adds u * |y| per step through the pole: <= 50 u max|y|."""
    e = e[:-1, :]
    E = np.empty_like(e)
    w = np.zeros(e.shape[0])
    for t in range(e.shape[1]):
        E[:, t] = e[:, t] + 0.02 * w
        w = e[:, t] + 0.98 * w
    E += emean
    if filt32:
        E += 2.0 * 50.0 * U * np.maximum(1.0, np.abs(y).max())
    return E


def locmax_guarded(v, ev):
    """(mask, near): O.locmax and, per bin, whether one of its two neighbour comparisons is inside the margin."""
    ge = np.zeros(len(v) + 1, dtype=bool)
    ge[0] = True
    ge[1:-1] = v[1:] >= v[:-1]
    mask = ge[:-1] & ~ge[1:]
    tie = np.zeros(len(v) + 1, dtype=bool)
    tie[1:-1] = np.abs(v[1:] - v[:-1]) <= ev[1:] + ev[:-1]
    near = tie[:-1] | tie[1:]
    return mask, near


def guarded_scan(sg, E, a_dec, G, K):
    """fwd_prune + bwd_prune of the oracle over `sg` with the guard; returns (peaks (P,2), flagged_frames set)."""
    srows, scols = sg.shape
    flagged = set()
    first = sg[:, :min(10, scols)]
    v0 = np.max(first, axis=1)
    thr = O.spreadpeaksinvector(v0, G)
    e0 = float(E[:, :min(10, scols)].max())
    terr = np.full(srows, e0)
    peaks = np.zeros((srows, scols))

    def bump(thr, terr, val, p, ev):
        nb = val * G[srows - p: 2 * srows - p]
        nerr = ev * G[srows - p: 2 * srows - p]
        close = np.abs(nb - thr) <= nerr + terr
        wins = nb > thr
        terr2 = np.where(wins, nerr, terr)
        terr2 = np.where(close, np.maximum(nerr, terr), terr2)
        return np.maximum(thr, nb), terr2
    for col in range(scols):
        s = sg[:, col]
        es = E[:, col]
        lm, near = locmax_guarded(s, es)
        above = s > thr
        marg = np.abs(s - thr) <= es + terr
        cand = lm & above
        # a comparison inside its margin that could change the candidate set: a local maximum near the threshold, or a bin that
        # is (or nearly is) above the threshold and whose local-maximum status hangs on a near-tie
        if col > 0 and (np.any(lm & marg) or np.any(near & (above | marg))):
            flagged.add(col)
        pos = np.nonzero(cand)[0]
        vp = sorted(zip(s[pos], pos), reverse=True)
        if len(vp) > K and abs(vp[K - 1][0] - vp[K][0]) <= es[vp[K - 1][1]] + es[vp[K][1]]:
            flagged.add(col)                                  # the top-K cut (:220-221)
        for val, p in vp[:K]:
            thr, terr = bump(thr, terr, val, p, es[p])
            peaks[p, col] = 1
        thr = thr * a_dec
        terr = terr * a_dec
    # backward
    last = sg[:, -1]
    thr = O.spreadpeaksinvector(last, G)
    terr = np.full(srows, float(E[:, -1].max()))
    for col in range(scols, 0, -1):
        pk = np.nonzero(peaks[:, col - 1])[0]
        for val, p in sorted(zip(sg[pk, col - 1], pk), reverse=True):
            if col < scols and abs(val - thr[p]) <= E[p, col - 1] + terr[p]:
                flagged.add(col - 1)                          # `val >= sthresh[peakpos]` (:242)
            if val >= thr[p]:
                thr, terr = bump(thr, terr, val, p, E[p, col - 1])
                if col < scols:
                    peaks[p, col] = 0
            else:
                peaks[p, col - 1] = 0
        thr = a_dec * thr
        terr = a_dec * terr
    cols, bins = np.nonzero(peaks.T)
    return np.stack([cols, bins], axis=1).astype(np.int32), flagged


def calibrate(clips):
    """Observed |log32 - log64| against the two terms of the bound: the smallest (c1, c2) that cover every value of `clips`."""
    r1, r2 = 0.0, 0.0
    for d in clips:
        S = O.stft_complex(d)
        mag = np.abs(S)
        if not mag.max() > 0:
            continue
        l64 = np.log(np.maximum(mag, mag.max() / 1e6))
        p32, nrm = spectral_f32(d)
        l32, _ = logs_f32(p32)
        live = mag > mag.max() / 1e5                           # (values at the floor are decided by the floor, not the FFT)
        err = np.abs(l32.astype(np.float64) - l64)[live] / U
        t1 = (nrm[None, :] / np.maximum(mag, 1e-300))[live]
        t2 = (1.0 + np.abs(l64))[live]
        # cover with c2 * t2 where t1 is small, then c1 from what remains
        small = t1 < np.percentile(t1, 20)
        r2 = max(r2, float((err[small] / t2[small]).max()))
        r1 = max(r1, float(np.maximum(err - r2 * t2, 0.0).max() / 1.0) if False else float(((err - r2 * t2) / t1).max()))
    return max(r1, 0.05), max(r2, 0.5)


def run_unit(d, prm, c1, c2, filt32):
    """one unit (clip x shift): dict(frames, flagged_frames, differs)"""
    st = O.find_peaks_stages(d, prm)
    T = 1 + len(d) // N_HOP
    if st.get('zero', False) or 'sgram' not in st:
        return dict(frames=T, flagged=0, differs=False, skipped=True)
    p32, nrm = spectral_f32(d)
    lg = logs_f32(p32)
    if lg is None:
        return dict(frames=T, flagged=0, differs=False, skipped=True)
    l32, fl = lg
    e = value_bound(p32, nrm, l32, c1, c2)
    # a value at (or within its bound of) the floor: the floor comparison itself is a decision -- give it the distance to the floor
    atfloor = p32 <= fl * np.float32(1.0 + 1e-5)
    e = np.where(atfloor, np.maximum(e, 4 * U * (1.0 + np.abs(l32))), e)
    mean64 = float(np.mean(l32.astype(np.float64)))
    emean = float(np.mean(e))
    if filt32:
        x = (l32 - np.float32(mean64)).astype(np.float32)
        y = hpf(x, np.float32).astype(np.float64)
    else:
        y = hpf(l32.astype(np.float64) - mean64, np.float64)
    E = filter_bound(e, emean, filt32, y)
    G = O.gauss_table(256, prm.f_sd)
    pk, flagged = guarded_scan(y, E, O.a_dec_of(prm.density, prm.n_hop), G, prm.maxpksperframe)
    differs = not np.array_equal(pk, st['peaks'])
    dmax = float(np.abs(y - st['sgram']).max())
    bound_ok = bool(np.all(np.abs(y - st['sgram']) <= E + 1e-12))
    return dict(frames=T, flagged=len(flagged), differs=differs, skipped=False, dmax=dmax, bound_ok=bound_ok,
                emed=float(np.median(E)), e99=float(np.percentile(E, 99)))


def workload(name, clips, prm, c1, c2, filt32):
    t0 = time.time()
    units = frames = fl_units = fl_frames = diff = diff_unflagged = bound_viol = 0
    dmax, e99 = 0.0, []
    for d in clips:
        for off in O.shift_offsets(prm.shifts):
            r = run_unit(d[off:], prm, c1, c2, filt32)
            units += 1
            frames += r['frames']
            if r['skipped']:
                continue
            fl_units += 1 if r['flagged'] else 0
            fl_frames += r['flagged']
            diff += 1 if r['differs'] else 0
            diff_unflagged += 1 if (r['differs'] and not r['flagged']) else 0
            bound_viol += 0 if r['bound_ok'] else 1
            dmax = max(dmax, r['dmax'])
            e99.append(r['e99'])
    out = dict(workload=name, filter='float32' if filt32 else 'float64', clips=len(clips), units=units, frames=frames,
               units_flagged=fl_units, units_flagged_frac=round(fl_units / max(1, units), 4),
               frames_flagged=fl_frames, frames_flagged_frac=round(fl_frames / max(1, frames), 6),
               units_whose_peaks_differ_from_f64=diff, differing_units_not_flagged=diff_unflagged,
               sound=bool(diff_unflagged == 0), units_where_a_value_left_its_bound=bound_viol,
               max_abs_sgram_error=dmax, bound_p99_median=float(np.median(e99)) if e99 else None, seconds=round(time.time() - t0, 1))
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=256)
    ap.add_argument('--c5-clips', type=int, default=32)
    ap.add_argument('--safety', type=float, default=2.0)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r06_f32_screen_feasibility.json'))
    a = ap.parse_args()
    cal = [O.synth_noise(99000 + i, 10.0) for i in range(8)] + [O.synth_tonal(99100 + i, 6.0) for i in range(4)]
    c1o, c2o = calibrate(cal)
    c1, c2 = a.safety * c1o, a.safety * c2o
    print('calibration: observed c1 %.3f c2 %.3f -> used c1 %.3f c2 %.3f (u = 2^-24)' % (c1o, c2o, c1, c2), flush=True)
    res = dict(what='float32 spectral stage + float64 guarded scan against the float64 oracle (tools/f32_screen_feasibility.py)',
               bound='e = u*(c1*||frame*window||/|S| + c2*(1+|log S|)), u=2^-24, through the onset filter (gain <= 2) and the mean',
               calibration=dict(observed_c1=c1o, observed_c2=c2o, safety=a.safety, c1=c1, c2=c2), rows=[])
    c3 = [O.synth_noise(1000003 * 0 + i, 30.0) for i in range(a.clips)]                 # bench.py's pool, rank 0
    p3 = O.Params()
    p5 = O.Params(density=70.0, maxpairsperpeak=10, shifts=4)
    for f32 in (False, True):
        res['rows'].append(workload('c3 noise %d x 30 s (density 20)' % a.clips, c3, p3, c1, c2, f32))
    # the same with the tightest bound that still covers the calibration set (safety 1: NOT a bound one could ship -- it is the
    # maximum error seen on 12 clips) and with half of that (unsound by construction; shows how the flagged share scales)
    for sf in (1.0, 0.5):
        r = workload('c3 noise %d x 30 s (density 20), bound = %.1f x the largest error observed in calibration' % (a.clips, sf),
                     c3, p3, sf * c1o, sf * c2o, False)
        r['safety'] = sf
        res['rows_other_safeties'] = res.get('rows_other_safeties', []) + [r]
    res['rows'].append(workload('c5 noise %d x 30 s (density 70, 4 shifts)' % a.c5_clips, c3[:a.c5_clips], p5, c1, c2, False))
    fx = {'tonal 16 x 20 s': [O.synth_tonal(5000 + i, 20.0) for i in range(16)],
          'fade (loud / quiet) 8': [O.synth_fade(6000 + i, lv) for i in range(4) for lv in (0.3, 0.003)],
          'noise-silence-noise 8': [np.concatenate([O.synth_noise(7000 + i, 4.0), np.zeros(22050, np.float32), O.synth_noise(7100 + i, 4.0)]) for i in range(8)]}
    for nm, cl in fx.items():
        res['rows'].append(workload(nm, cl, p3, c1, c2, False))
    c3row = res['rows'][0]
    res['kill_criterion'] = '> 15 % of C3 units flagged'
    res['c3_units_flagged_frac'] = c3row['units_flagged_frac']
    res['all_sound'] = bool(all(r['sound'] for r in res['rows']))
    res['decision'] = ('KILLED: %.1f %% of C3 units would be re-run in float64' % (100 * c3row['units_flagged_frac'])
                       if c3row['units_flagged_frac'] > 0.15 else 'survives the kill criterion')
    with open(a.out, 'w') as f:
        json.dump(res, f, indent=1)
    print('decision:', res['decision'], '| sound:', res['all_sound'])


if __name__ == '__main__':
    main()
