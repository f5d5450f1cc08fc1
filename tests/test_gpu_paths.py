"""GPU: the three kernel paths of the extraction -- dense (k_stft -> k_scan), COMPACT (k_stft<ST,true> -> k_scan_c: the
log-spectrogram never reaches HBM) and SEGMENT-parallel scan (k_hpf -> k_scan_seg) -- must give the reference's integers on
every input; forced through afp_set_pipeline so each is exercised whatever the batch size."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

PKEYS = ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts', 'targetdf', 'mindt', 'targetdt')
PATHS = {'dense': dict(compact=0, seg=0), 'compact': dict(compact=1, seg=0), 'segments': dict(compact=0, seg=1),
         'segments_short_warmup': dict(compact=0, seg=1, seg_len=16, seg_warm=4)}


@pytest.fixture(scope='module')
def ex():
    from audfprint_amd.batch import Extractor
    e = Extractor.get(0)
    yield e
    e.set_pipeline()            # defaults back for the other test modules


def _params_of(g):
    p = dict(g['params'])
    return p


@pytest.mark.parametrize('path', sorted(PATHS))
def test_every_golden_on_every_path(ex, path):
    """All golden fixtures (generated from the live reference) as ONE mixed batch per parameter set, on a forced path."""
    from audfprint_amd import _lib
    ex.set_pipeline(**PATHS[path])
    groups = {}
    for name in golden_names():
        g = load_golden(name)
        groups.setdefault(tuple(sorted((k, g['params'][k]) for k in PKEYS)), []).append((name, g))
    assert groups
    for key, items in groups.items():
        ex.set_params(**dict(key))
        r = ex.extract(clips=[g['d'] for _, g in items], want_hashes=True, want_peaks=True)
        for i, (name, g) in enumerate(items):
            tie = bool(r.unit_flags[i * len(g['peaks'])] & _lib.UNIT_TIE)
            if tie:
                continue        # the lone-click class (tests/test_gpu_parity.py checks its own contract)
            for sft, pk in enumerate(g['peaks']):
                assert np.array_equal(r.unit_peaks(i, sft), pk), (path, name, sft)
            assert np.array_equal(r.clip_hashes(i), g['hashes']), (path, name)


def test_long_clips_segmented_vs_oracle(ex):
    """300 s of a tonal + gated signal and a 3600 s clip: segments (default parameters) against the CPU oracle; no segment
    re-run is expected, the final boundary check must pass."""
    from oracle import afp_oracle as O
    ex.set_pipeline(compact=0, seg=1)
    ex.set_params()
    for d in (O.synth_tonal(31, 300.0), np.tile(O.synth_noise(32, 600.0), 6)):
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['segments'] > 20 and not st['failed'], st
        pls, hs = O.extract(d, O.Params())
        assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs)


def test_segment_repair_and_fallback_are_exact(ex):
    """A warm-up far too short to converge: the chain launch re-runs the segments (whole runs of them, sequentially, from the
    last true state); with the final check forced to fail the sequential kernel produces the result -- bit-exact either way."""
    from oracle import afp_oracle as O
    ex.set_params()
    d = O.synth_noise(41, 120.0)
    pls, hs = O.extract(d, O.Params())
    seen_rerun = False
    for seg_len, warm, force in ((64, 8, False), (256, 32, False), (128, 2, False), (128, 2, True), (0, 0, True)):
        ex.set_pipeline(compact=0, seg=1, seg_len=seg_len, seg_warm=warm, seg_force_fail=force)
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['failed'] == force, st
        seen_rerun = seen_rerun or st['rerun_fwd'] + st['rerun_bwd'] > 0
        assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs), (seg_len, warm, st)
    assert seen_rerun


def test_compact_equals_dense_on_a_ragged_batch(ex):
    """1100 clips of 0.02 .. 12 s (noise, tonal, silence, clipped): compact and dense paths row for row, units that need
    the floor included (they take the dense kernels inside the compact pipeline)."""
    from oracle import afp_oracle as O
    from audfprint_amd import _lib
    rng = np.random.RandomState(7)
    clips = []
    for i in range(1100):
        secs = float(rng.uniform(0.02, 12.0))
        kind = i % 5
        if kind == 0:
            d = O.synth_tonal(2000 + i, max(secs, 0.5))[:int(secs * 11025) + 1]
        elif kind == 1:
            d = np.zeros(int(secs * 11025), np.float32)
        elif kind == 2:
            d = np.clip(O.synth_noise(2000 + i, secs) * 20.0, -1.0, 1.0).astype(np.float32)
        else:
            d = O.synth_noise(2000 + i, secs)
        clips.append(d)
    ex.set_params()
    ex.set_pipeline(compact=0, seg=0)
    a = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    ex.set_pipeline(compact=1, seg=0)
    b = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    assert np.array_equal(a.hash_offsets, b.hash_offsets) and np.array_equal(a.hashes, b.hashes)
    assert np.array_equal(a.peak_offsets, b.peak_offsets) and np.array_equal(a.peaks, b.peaks)
    assert np.array_equal(a.unit_flags, b.unit_flags)
    assert int(np.count_nonzero(a.unit_flags & _lib.UNIT_CORR)) > 0          # the floored path was part of it
    for i in (3, 4, 8, 9, 503, 1099):
        pls, hs = O.extract(clips[i], O.Params())
        assert np.array_equal(b.clip_hashes(i), hs) and np.array_equal(b.unit_peaks(i), pls[0]), i


def _gappy(seed, secs=120.0):
    """Noise with a loud burst, a stretch 30 dB down and 6 s of digital silence: the thresholds of :226-230 remember the
    loud part for hundreds of frames, so segments that start in the quiet stretches cannot converge from their warm-up."""
    from oracle import afp_oracle as O
    d = O.synth_noise(seed, secs).copy()
    sr = 11025
    d[10 * sr:12 * sr] *= 8.0
    d[12 * sr:40 * sr] *= 0.03
    d[60 * sr:66 * sr] = 0.0
    d[66 * sr:67 * sr] *= 6.0
    d[67 * sr:90 * sr] *= 0.1
    return np.clip(d, -1.0, 1.0).astype(np.float32)


def test_segments_over_quiet_stretches(ex):
    """Runs of segments whose warm-up cannot converge: the chain launch re-runs each run sequentially from the last true
    state -- the oracle's result, and no unit left to the sequential kernel."""
    from oracle import afp_oracle as O
    ex.set_params()
    for seed, dens in ((51, 20.0), (52, 70.0)):
        ex.set_params(density=dens)
        d = _gappy(seed)
        pls, hs = O.extract(d, O.Params(density=dens))
        ex.set_pipeline(compact=0, seg=1)
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['segments'] > 20, st
        assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs), st
        assert st['rerun_fwd'] + st['rerun_bwd'] > 0, st                  # the signal does break the warm-up premise
        assert not st['failed'], st
    ex.set_params()


def test_segments_in_a_batch_of_long_clips(ex):
    """Several long clips in one batch (some with quiet stretches), segmented: all equal the dense path's; then with the final
    check forced to fail: every unit re-done by the sequential kernel, same result."""
    from oracle import afp_oracle as O
    ex.set_params()
    clips = [O.synth_noise(61, 60.0), _gappy(62, 100.0), O.synth_tonal(63, 45.0), _gappy(64, 95.0), O.synth_noise(65, 30.0)]
    ex.set_pipeline(compact=0, seg=0)
    r0 = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    h0, p0, o0 = r0.hashes.copy(), r0.peaks.copy(), r0.hash_offsets.copy()
    for force in (False, True):
        ex.set_pipeline(compact=0, seg=1, seg_force_fail=force)
        r1 = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['failed_units'] == (len(clips) if force else 0), st
        assert np.array_equal(h0, r1.hashes) and np.array_equal(p0, r1.peaks) and np.array_equal(o0, r1.hash_offsets), st
