"""GPU: input corners the reference handles without raising -- NaN / Inf samples, and sparse frames (a lone click and the rest of
the class whose peaks are FFT rounding noise in the reference itself)."""
import numpy as np
import pytest

from conftest import SPARSE_CONTROLS, SPARSE_FRAME, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ex():
    from audfprint_amd.batch import Extractor
    e = Extractor.get(0)
    yield e
    e.set_pipeline()


@pytest.mark.parametrize('path', ['dense', 'compact', 'segments'])
def test_nan_and_inf_samples_take_the_reference_zero_branch(ex, path):
    """np.max over a spectrogram holding a NaN is NaN, `smax > 0` is false: the reference prints the "identically zero"
    warning and finds no peaks (audfprint_analyze.py:283-290) -- for a NaN and, through inf - inf, for an Inf sample.  The
    oracle does the same (verified against the live reference); the GPU must agree, per unit, and leave the other units
    of the batch untouched."""
    from oracle import afp_oracle as O
    from audfprint_amd import _lib
    ex.set_pipeline(**{'dense': dict(compact=0, seg=0), 'compact': dict(compact=1, seg=0), 'segments': dict(compact=0, seg=1)}[path])
    ex.set_params()
    base = O.synth_noise(5, 30.0)
    clips, bad = [], []
    for k, (pos, val) in enumerate([(12000, np.nan), (12000, np.inf), (12000, -np.inf), (0, np.nan), (len(base) - 1, np.inf),
                                    (None, None), (70000, np.nan), (None, None)]):
        d = base.copy() if k % 2 == 0 else O.synth_noise(50 + k, 30.0)
        if pos is not None:
            d[pos] = val
        clips.append(d)
        bad.append(pos is not None)
    r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    for i, d in enumerate(clips):
        with np.errstate(all='ignore'):
            pls, hs = O.extract(d, O.Params())
        if bad[i]:
            # NaN: exactly the reference (no peaks).  +-Inf: the reference's own result is an accident of where inf - inf
            # turns into NaN inside pocketfft / np.abs (an Inf in the last frame leaves it one "peak" at bin 0); the
            # library's contract is: no peaks, unit flagged.
            assert len(r.unit_peaks(i)) == 0 and len(r.clip_hashes(i)) == 0, i
            assert (r.unit_flags[i] & _lib.UNIT_ZERO) and (r.unit_flags[i] & _lib.UNIT_NONFINITE), i
            if np.isnan(d).any():
                assert len(pls[0]) == 0 and len(hs) == 0
        else:
            assert np.array_equal(r.unit_peaks(i), pls[0]) and np.array_equal(r.clip_hashes(i), hs), i
            assert len(hs) > 0 and not (r.unit_flags[i] & (_lib.UNIT_ZERO | _lib.UNIT_NONFINITE))


def test_a_huge_finite_sample_is_not_a_nan(ex):
    """A finite click 1e7 times the noise around it is ordinary input (only a click so large that the noise vanishes in
    float64 next to it -- 1e38 -- degenerates into the lone-click class, where FFT rounding noise picks the bins)."""
    from oracle import afp_oracle as O
    ex.set_pipeline()
    ex.set_params()
    d = O.synth_noise(5, 3.0)
    d[12000] = 1e6
    r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
    pls, hs = O.extract(d, O.Params())
    assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs) and len(pls[0]) > 0


def test_tie_frames_bound_the_divergence_of_a_lone_click(ex):
    """hand_impulse (one non-zero sample in digital silence): the unit is flagged AFP_UNIT_TIE and afp_fetch_unit_tie_frames
    names the frames whose spectrum is flat to the last bit.  Every peak of the reference AND of the GPU lies in those
    frames; outside them the two peak lists are identical (here: empty)."""
    from audfprint_amd import _lib
    g = load_golden('hand_impulse')
    ex.set_pipeline()
    ex.set_params(**{k: g['params'][k] for k in ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts', 'targetdf', 'mindt', 'targetdt')})
    # a second, ordinary clip in the same batch keeps its own (empty) range
    from oracle import afp_oracle as O
    r = ex.extract(clips=[g['d'], O.synth_noise(3, 2.0)], want_hashes=True, want_peaks=True)
    first, last = ex.tie_frames()
    assert r.unit_flags[0] & _lib.UNIT_TIE and not (r.unit_flags[1] & _lib.UNIT_TIE)
    assert first[1] == 0 and last[1] == -1
    nz = np.flatnonzero(g['d'])
    assert len(nz) == 1
    # frames whose 512-sample window (256 t - 256 .. 256 t + 255) holds the click
    t_lo, t_hi = int(nz[0] + 256 - 511 + 255) // 256, int(nz[0] + 256) // 256
    assert first[0] == t_lo and last[0] == t_hi, (first[0], last[0], t_lo, t_hi)
    ref = g['peaks'][0]
    got = r.unit_peaks(0)
    assert len(ref) > 0 and np.all((ref[:, 0] >= first[0]) & (ref[:, 0] <= last[0]))
    inside = (got[:, 0] >= first[0]) & (got[:, 0] <= last[0])
    assert np.array_equal(got[~inside], ref[(ref[:, 0] < first[0]) | (ref[:, 0] > last[0])])


@pytest.mark.parametrize('name', [n for n in SPARSE_FRAME if n != 'hand_impulse'])
@pytest.mark.parametrize('path', ['dense', 'compact', 'segments'])
def test_sparse_frames_on_a_signal_that_continues(ex, name, path):
    """the sparse-frame class (all non-zero samples of a frame at offsets of one parity: a lone
    click, two clicks 64 / 100 / 128 / 256 apart with equal and unequal amplitudes, three at spacing 128, four at spacing 64)
    in 1 s of digital silence followed by 4 s of noise at -50 dB and at -20 dB, and the +-1 LSB tail of an undithered
    fade-out -- fixtures from the live reference.  The contract of include/afp.h as numbers, on every kernel path: the unit is
    flagged; the tie range covers every frame the rule names (oracle.sparse_parity_frames, a numpy restatement of the
    detector) and nothing outside the hull of the single-parity frames; the peak lists are IDENTICAL in every frame more than
    two decay lengths away from the range, and -- where silence precedes the clicks -- in every frame before it.  The counts
    land in gpurun_out/sparse_frame_divergence.json (profiles/r05_sparse_frame_divergence.json is a committed copy)."""
    import json
    import os
    from audfprint_amd import _lib
    from oracle import afp_oracle as O
    g = load_golden(name)
    ex.set_pipeline(**dict(dense=dict(compact=0, seg=0), compact=dict(compact=1, seg=0), segments=dict(compact=0, seg=1, seg_len=16, seg_warm=32))[path])
    ex.set_params(**{k: g['params'][k] for k in ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts', 'targetdf', 'mindt', 'targetdt')})
    try:
        r = ex.extract(clips=[g['d']], want_hashes=True, want_peaks=True)
        first, last = ex.tie_frames()
    finally:
        ex.set_pipeline()
    assert r.unit_flags[0] & _lib.UNIT_TIE
    above, every, _bound, _floor = O.sparse_parity_frames(g['d'])
    assert above, 'the fixture must hold a single-parity frame above the floor'
    assert first[0] <= min(above) and last[0] >= max(above), (first[0], last[0], above)
    assert first[0] >= min(every) and last[0] <= max(every), (first[0], last[0], every)
    clicks = name != 'fade_quiet'
    if clicks:
        # every single-parity frame of these fixtures holds a click of amplitude >= 0.25: the range is exactly their hull
        nz = np.flatnonzero(g['d'][:11025])
        assert (first[0], last[0]) == ((int(nz[0]) + 256 - 511 + 255) // 256, (int(nz[-1]) + 256) // 256), (first[0], last[0], nz)
    ref, got = g['peaks'][0], r.unit_peaks(0)
    T = 1 + len(g['d']) // 256
    def per_frame(p):
        return [tuple(p[p[:, 0] == t, 1].tolist()) for t in range(T)]
    fr, fg = per_frame(ref), per_frame(got)
    differ = [t for t in range(T) if fr[t] != fg[t]]
    before = [t for t in differ if t < first[0]]
    inside = [t for t in differ if first[0] <= t <= last[0]]
    after = [t for t in differ if t > last[0]]
    rec = dict(fixture=name, path=path, tie_frames=[int(first[0]), int(last[0])], frames=T, ref_peaks=int(len(ref)), gpu_peaks=int(len(got)),
               frames_differing_before=len(before), frames_differing_inside=len(inside), frames_differing_after=len(after),
               first_differing_frame=(min(differ) if differ else None), last_differing_frame=(max(differ) if differ else None),
               peaks_differing=int(len(set(map(tuple, ref.tolist())) ^ set(map(tuple, got.tolist())))),
               hashes_equal=bool(np.array_equal(r.clip_hashes(0), g['hashes'])))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'sparse_frame_divergence.json'), 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
    print(rec)
    a_dec = (1 - 0.01 * (g['params']['density'] * np.sqrt(256 / 352.8) / 35))
    reach = int(2.0 / (1 - a_dec))
    if clicks:
        assert not before, rec                  # the backward pass runs from the end: it could carry a difference to earlier frames
                                                # only through a threshold raised inside the range, and the silence before holds no peak
    assert all(t >= first[0] - reach for t in before), rec
    assert all(t <= last[0] + reach for t in after), rec


@pytest.mark.parametrize('path', ['dense', 'compact', 'segments'])
def test_sparse_frames_with_both_parities_are_exact_and_unflagged(ex, path):
    """The controls: two clicks 101 samples apart, four clicks 63 apart (each frame that holds any of them holds both
    parities) and a LOUD undithered fade-out (its last non-silent frames are dense).  The live reference does not move on
    these under FFT jitter (profiles/r05_sparse_frame_jitter_reference.json), so the library must be bit-exact and must not
    raise the flag -- on every kernel path."""
    from audfprint_amd import _lib
    ex.set_pipeline(**dict(dense=dict(compact=0, seg=0), compact=dict(compact=1, seg=0), segments=dict(compact=0, seg=1, seg_len=16, seg_warm=32))[path])
    try:
        for name in SPARSE_CONTROLS:
            g = load_golden(name)
            ex.set_params(**{k: g['params'][k] for k in ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts', 'targetdf', 'mindt', 'targetdt')})
            r = ex.extract(clips=[g['d']], want_hashes=True, want_peaks=True)
            assert not (r.unit_flags[0] & _lib.UNIT_TIE), (name, path)
            assert np.array_equal(r.unit_peaks(0), g['peaks'][0]) and np.array_equal(r.clip_hashes(0), g['hashes']), (name, path)
    finally:
        ex.set_pipeline()


def test_unusual_sample_values_equal_the_oracle():
    """Sample values a decoder normally never hands over but the arithmetic must still follow the reference on: tiny and huge
    gains, a DC offset, float32 denormals, a full-scale square wave, the int16 extremes through the s16 entry, one huge spike
    in quiet noise (drives most of the clip under the floor max|S| / 1e6, audfprint_analyze.py:285) -- all three kernel paths."""
    from audfprint_amd.batch import Extractor
    from oracle import afp_oracle as O
    ex = Extractor.get(0)
    ex.set_params()
    rng = np.random.RandomState(77)
    base = O.synth_noise(4242, 6.0)
    n = len(base)
    spike = (base * np.float32(1e-3)).copy()
    spike[n // 2] = 0.9
    den = (base * np.float32(1e-6)).copy()
    den[100:4000] = np.float32(1e-40) * np.sign(base[100:4000])          # float32 denormals
    sq = np.where((np.arange(n) // 37) % 2 == 0, 1.0, -1.0).astype(np.float32)
    clips = [base * np.float32(1e-6), base * np.float32(1e4), (base * np.float32(0.3) + np.float32(0.5)).astype(np.float32),
             den, sq, spike]
    prm = O.Params()
    want = [O.extract(d, prm) for d in clips]
    try:
        for name, kw in (('dense', dict(compact=0, seg=0)), ('compact', dict(compact=1, seg=0)), ('segments', dict(compact=0, seg=1))):
            ex.set_pipeline(**kw)
            r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
            for i, (pls, hs) in enumerate(want):
                assert np.array_equal(r.unit_peaks(i), pls[0]) and np.array_equal(r.clip_hashes(i), hs), (name, i)
        # the int16 extremes: -32768 and 32767 next to each other, through the raw s16 entry (audio_read.py:121-145: / 32768)
        s16 = rng.randint(-3000, 3000, n).astype(np.int16)
        s16[::997] = -32768
        s16[1::997] = 32767
        ex.set_pipeline()
        r = ex.extract(clips=[s16], want_hashes=True, want_peaks=True)
        pls, hs = O.extract(s16.astype(np.float32) / np.float32(32768), prm)
        assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs)
    finally:
        ex.set_pipeline()
