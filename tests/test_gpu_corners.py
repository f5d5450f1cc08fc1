"""GPU: input corners the reference handles without raising -- NaN / Inf samples, and the frames of a lone click."""
import numpy as np
import pytest

from conftest import load_golden

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


@pytest.mark.parametrize('name', ['hand_click_then_noise', 'hand_click_then_quiet_noise'])
@pytest.mark.parametrize('path', ['dense', 'compact', 'segments'])
def test_lone_click_on_a_signal_that_continues(ex, name, path):
    """a lone click in 1 s of digital silence followed by 4 s of noise (fixtures from the live reference).  The
    contract of INTEGRATION.md as numbers: the unit is flagged, the tie range is the two frames that hold the click, the peak
    lists are IDENTICAL in every frame before the range, and the frames after it that differ are counted -- the count lands
    in gpurun_out/lone_click_divergence.json (profiles/r04_lone_click_divergence.json is a committed copy) and is bounded by
    what the decaying threshold can remember (2 decay lengths)."""
    import json
    import os
    from audfprint_amd import _lib
    g = load_golden(name)
    ex.set_pipeline(**dict(dense=dict(compact=0, seg=0), compact=dict(compact=1, seg=0), segments=dict(compact=0, seg=1, seg_len=16, seg_warm=32))[path])
    ex.set_params(**{k: g['params'][k] for k in ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts', 'targetdf', 'mindt', 'targetdt')})
    try:
        r = ex.extract(clips=[g['d']], want_hashes=True, want_peaks=True)
        first, last = ex.tie_frames()
    finally:
        ex.set_pipeline()
    assert r.unit_flags[0] & _lib.UNIT_TIE
    nz = 5000                                   # the click (tests/golden/make_golden.py)
    assert g['d'][nz] != 0 and not np.any(g['d'][:nz]) and not np.any(g['d'][nz + 1:11025])
    assert (first[0], last[0]) == ((nz + 256 - 511 + 255) // 256, (nz + 256) // 256)
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
               last_differing_frame=(max(differ) if differ else None),
               hashes_equal=bool(np.array_equal(r.clip_hashes(0), g['hashes'])))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'lone_click_divergence.json'), 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
    print(rec)
    assert not before, rec                      # the backward pass runs from the end: it could carry a difference to earlier frames
                                                # only through a threshold raised inside the range, and the silence before holds no peak
    a_dec = (1 - 0.01 * (g['params']['density'] * np.sqrt(256 / 352.8) / 35))
    assert all(t <= last[0] + int(2.0 / (1 - a_dec)) for t in after), rec
