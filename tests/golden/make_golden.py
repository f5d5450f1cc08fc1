#!/usr/bin/env python
"""Generate the golden fixtures in this directory FROM THE LIVE REFERENCE.

Run in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py

It imports the reference's own ``audfprint_analyze`` / ``stft`` modules (unchanged) and
records, per case, the input recipe (or the raw int16 PCM for hand-made signals), the
Analyzer parameters, the per-shift peak lists from ``Analyzer.find_peaks``
(audfprint_analyze.py:255-308) and the sorted-unique hashes from the
``wavfile2hashes`` logic (audfprint_analyze.py:404-422).  A few small cases also keep the
float intermediates (|S|, mean-subtracted log spectrogram, HPF'd spectrogram) for the
1e-4 float-parity check.  The reference has no golden vectors of its own
(SURVEY.md §8c); these files are what pins the oracle and the HIP path.

The GPU box has no /root/reference: tests only read the .npz files written here.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(HERE, '..', '..'))

import audfprint_analyze as REF  # noqa: E402  (the reference, unchanged)
import stft as REFSTFT  # noqa: E402
import scipy.signal  # noqa: E402
from oracle import afp_oracle as O  # noqa: E402  (only for the synthetic-input recipes)

DEFAULTS = dict(density=20.0, maxpksperframe=5, maxpairsperpeak=3, f_sd=30.0, shifts=1,
                targetdf=31, mindt=2, targetdt=63)


def make_input(spec):
    kind = spec['kind']
    if kind == 'noise':
        return O.synth_noise(spec['seed'], spec['secs'], nsamp=spec.get('nsamp'))
    if kind == 'tonal':
        return O.synth_tonal(spec['seed'], spec['secs'])
    if kind == 'clicks':
        return O.synth_clicks(spec['pos'], spec['amp'], spec['tail_db'])
    if kind == 'fade':
        return O.synth_fade(spec['seed'], spec['level'])
    raise ValueError(kind)


def handmade(name):
    """Hand-made int16 signals (stored raw in the fixture)."""
    rng = np.random.RandomState(1234)
    if name == 'zeros_1s':
        return np.zeros(11025, np.int16)
    if name == 'silence_then_noise':
        x = np.concatenate([np.zeros(2 * 11025), rng.randn(3 * 11025) * 0.1])
        return np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    if name == 'noise_silence_noise':
        x = np.concatenate([rng.randn(11025) * 0.2, np.zeros(11025 + 77), rng.randn(11025) * 0.05])
        return np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    if name == 'clipped':
        x = rng.randn(4 * 11025) * 2.0
        return np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    if name == 'impulse':
        x = np.zeros(2 * 11025)
        x[11025] = 1.0
        return np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    if name == 'click_then_noise':
        # one sample at 0.5 in 1 s of digital silence, then 4 s of noise -- the lone-click class (AFP_UNIT_TIE)
        # on a signal that CONTINUES after the click, so that the frames after the tie range hold reference peaks
        x = np.concatenate([np.zeros(11025), rng.randn(4 * 11025) * 0.1])
        x[5000] = 0.5
        return np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    if name == 'click_then_quiet_noise':
        # the same with noise 44 dB under the click: here the click frames KEEP reference peaks through the backward pass, and
        # the thresholds they raise reach into the noise that follows
        x = np.concatenate([np.zeros(11025), rng.randn(4 * 11025) * 0.003])
        x[5000] = 0.5
        return np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    if name == 'dc_step':
        x = np.zeros(3 * 11025)
        x[5000:] = 0.25
        return np.round(x * 32767).astype(np.int16)
    if name == 'sine_fullscale':
        t = np.arange(3 * 11025) / 11025.0
        return np.round(np.sin(2 * np.pi * 1000.0 * t) * 32767).astype(np.int16)
    raise ValueError(name)


def run_reference(d, prm):
    an = REF.Analyzer(prm['density'])
    an.maxpksperframe = prm['maxpksperframe']
    an.maxpairsperpeak = prm['maxpairsperpeak']
    an.f_sd = prm['f_sd']
    an.shifts = prm['shifts']
    an.targetdf = prm['targetdf']
    an.mindt = prm['mindt']
    an.targetdt = prm['targetdt']
    # wavfile2peaks shifts loop, audfprint_analyze.py:369-377
    if prm['shifts'] < 2:
        peaklists = [an.find_peaks(d, 11025)]
    else:
        peaklists = [an.find_peaks(d[int(s / an.shifts * an.n_hop):], 11025)
                     for s in range(prm['shifts'])]
    # wavfile2hashes, audfprint_analyze.py:404-422
    qh = np.concatenate([REF.landmarks2hashes(an.peaks2landmarks(p)) for p in peaklists])
    if qh.shape[0]:
        hh = ((qh[:, 0].astype(np.uint64)) << np.uint64(32)) + qh[:, 1].astype(np.uint64)
        u = np.sort(np.unique(hh))
        hashes = np.hstack([(u >> np.uint64(32))[:, None],
                            (u & np.uint64((1 << 32) - 1))[:, None]]).astype(np.int32)
    else:
        hashes = np.zeros((0, 2), np.int32)
    landmarks0 = np.array(an.peaks2landmarks(peaklists[0]), dtype=np.int32).reshape(-1, 4)
    return an, peaklists, landmarks0, hashes


def stages_reference(d, an):
    """Float intermediates exactly as Analyzer.find_peaks computes them (:277-295)."""
    mywin = np.hanning(an.n_fft + 2)[1:-1]
    S = REFSTFT.stft(d, n_fft=an.n_fft, hop_length=an.n_hop, window=mywin)
    mag = np.abs(S)
    sg = mag
    if np.max(sg) > 0.0:
        sg = np.log(np.maximum(sg, np.max(sg) / 1e6))
        sg = sg - np.mean(sg)
    hp = np.array([scipy.signal.lfilter([1, -1], [1, -REF.HPF_POLE ** (1 / REF.OVERSAMP)], r)
                   for r in sg])[:-1, ]
    a_dec = (1 - 0.01 * (an.density * np.sqrt(an.n_hop / 352.8) / 35)) ** (1 / REF.OVERSAMP)
    fwd = an._decaying_threshold_fwd_prune(hp, a_dec)
    return mag, sg, hp, fwd


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


CASES = [
    # name, input spec, param overrides, keep float stages?
    ('noise_s0_10s', dict(kind='noise', seed=0, secs=10), {}, False),
    ('noise_s1_10s', dict(kind='noise', seed=1, secs=10), {}, False),
    ('noise_s0_30s', dict(kind='noise', seed=0, secs=30), {}, False),
    ('noise_s0_300s', dict(kind='noise', seed=0, secs=300), {}, False),
    ('noise_s0_30s_c5', dict(kind='noise', seed=0, secs=30), dict(density=70.0, maxpairsperpeak=10, shifts=4), False),
    ('noise_s2_2s_stages', dict(kind='noise', seed=2, secs=2), {}, True),
    ('tonal_s3_20s', dict(kind='tonal', seed=3, secs=20), {}, False),
    ('tonal_s4_12s_c5', dict(kind='tonal', seed=4, secs=12), dict(density=70.0, maxpairsperpeak=10, shifts=4), False),
    ('tonal_s5_3s_stages', dict(kind='tonal', seed=5, secs=3), {}, True),
    ('noise_s7_8s_k2', dict(kind='noise', seed=7, secs=8), dict(maxpksperframe=2, density=100.0), False),
    ('noise_s8_8s_k8', dict(kind='noise', seed=8, secs=8), dict(maxpksperframe=8, density=200.0, maxpairsperpeak=6), False),
    ('noise_s9_8s_fsd15', dict(kind='noise', seed=9, secs=8), dict(f_sd=15.0, density=40.0), False),
    ('noise_s10_8s_pairgeom', dict(kind='noise', seed=10, secs=8), dict(targetdf=10, mindt=1, targetdt=30, maxpairsperpeak=5), False),
    ('noise_s11_8s_sh2', dict(kind='noise', seed=11, secs=8), dict(shifts=2), False),
    ('noise_s12_8s_sh3', dict(kind='noise', seed=12, secs=8), dict(shifts=3, density=50.0, maxpairsperpeak=8), False),
    ('tonal_s13_10s_d200', dict(kind='tonal', seed=13, secs=10), dict(density=200.0, maxpairsperpeak=10), False),
] + [
    ('noise_n%d' % n, dict(kind='noise', seed=100 + n, secs=0, nsamp=n), {}, False)
    for n in (1, 2, 100, 255, 256, 257, 511, 512, 513, 767, 768, 2000)
] + [
    ('hand_' + nm, dict(kind='hand', name=nm), {}, nm in ('silence_then_noise',))
    for nm in ('zeros_1s', 'silence_then_noise', 'noise_silence_noise', 'clipped', 'impulse',
               'dc_step', 'sine_fullscale', 'click_then_noise', 'click_then_quiet_noise')
] + [
    # the sparse-frame class beyond the lone click -- frames whose non-zero samples share one parity
    # (two clicks an even distance apart, equal and 0.5 / 0.25; three at spacing 128; four at spacing 64), in 1 s of digital
    # silence, followed by noise at -50 dB (the click frames keep reference peaks) and at -20 dB (the backward pass prunes them)
    ('sparse_%s_%ddb' % (nm, -db), dict(kind='clicks', pos=pos, amp=amp, tail_db=float(db)), {}, False)
    for db in (-50, -20)
    for nm, pos, amp in
    [('two_d%d_%s' % (dl, an), [5000, 5000 + dl], am) for dl in (64, 100, 128, 256) for an, am in (('eq', [0.5, 0.5]), ('uneq', [0.5, 0.25]))]
    + [('three_s128', [5000, 5128, 5256], [0.5, 0.5, 0.5]), ('four_s64', [5000, 5064, 5128, 5192], [0.5, 0.5, 0.5, 0.5]),
       # controls that must stay UNFLAGGED and bit-exact: two clicks an odd distance apart inside one frame
       ('two_d101_eq', [5000, 5101], [0.5, 0.5]), ('four_s63', [5000, 5063, 5126, 5189], [0.5, 0.5, 0.5, 0.5])]
] + [
    # an undithered fade-out: noise at -10 dBFS / -80 dBFS, linear fade to digital silence, int16
    ('fade_loud', dict(kind='fade', seed=5, level=0.3), {}, False),
    ('fade_quiet', dict(kind='fade', seed=5, level=1e-4), {}, False),
] + [
    ('hand_silence_then_noise_c5', dict(kind='hand', name='silence_then_noise'),
     dict(density=70.0, maxpairsperpeak=10, shifts=4), False),
]


def main():
    index = {}
    only = [a for a in sys.argv[1:] if not a.startswith('-')]       # optional: regenerate just the named cases
    if only:
        with open(os.path.join(HERE, 'INDEX.json')) as f:
            index = json.load(f)
    for name, spec, over, keep in CASES:
        if only and name not in only:
            continue
        prm = dict(DEFAULTS)
        prm.update(over)
        payload = {}
        if spec['kind'] == 'hand':
            pcm = handmade(spec['name'])
            payload['pcm_i16'] = pcm
            d = pcm.astype(np.float32) / np.float32(32768)      # audio_read.buf_to_float :121-145
        else:
            d = make_input(spec)
        an, peaklists, lm0, hashes = run_reference(d, prm)
        payload['meta'] = np.array(json.dumps(dict(spec=spec, params=prm, nsamp=int(len(d)),
                                                   pcm_sha=sha(d))))
        for s, p in enumerate(peaklists):
            payload['peaks%d' % s] = np.array(p, dtype=np.int32).reshape(-1, 2)
        payload['landmarks0'] = lm0
        payload['hashes'] = hashes
        if keep:
            mag, sg, hp, fwd = stages_reference(d, an)
            payload['mag'] = mag
            payload['logs'] = sg
            payload['sgram'] = hp
            payload['fwd'] = np.packbits(fwd.astype(bool), axis=0)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **payload)
        index[name] = dict(nsamp=int(len(d)), npeaks=[int(len(p)) for p in peaklists],
                           nhashes=int(len(hashes)), peaks_sha=sha(payload['peaks0']),
                           hashes_sha=sha(hashes), pcm_sha=sha(d))
        print(name, index[name])
    with open(os.path.join(HERE, 'INDEX.json'), 'w') as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
