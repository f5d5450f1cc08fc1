"""GPU: the drop-in Analyzer (audfprint_amd.audfprint_analyze) used the way audfprint.py uses the
reference class -- file in, hashes / peaks / table insert out -- against the golden fixtures."""
import os
import sys
import types

import numpy as np
import pytest
import scipy.io.wavfile

from conftest import load_golden

pytestmark = pytest.mark.gpu

import audfprint_amd.audfprint_analyze as M  # noqa: E402


@pytest.fixture(autouse=True)
def fake_audio_read(monkeypatch):
    """Stand-in for the reference's audio_read module (ffmpeg is not installed anywhere here):
    11025 Hz mono s16 WAV -> float32/32768, exactly audio_read.buf_to_float (audio_read.py:121-145)."""
    mod = types.ModuleType('audio_read')

    def audio_read(filename, sr=None, channels=None):
        rate, w = scipy.io.wavfile.read(filename)
        if sr and sr != rate:
            raise ValueError('samplerate')
        return w.astype(np.float32) / np.float32(32768), rate
    mod.audio_read = audio_read
    monkeypatch.setitem(sys.modules, 'audio_read', mod)


def _write_wav(path, d):
    scipy.io.wavfile.write(path, 11025, np.round(d * 32768).astype(np.int16))


def _setup(a, p):
    # what audfprint.py:setup_analyzer does (audfprint.py:280-299)
    a.density, a.maxpksperframe, a.maxpairsperpeak = p['density'], p['maxpksperframe'], p['maxpairsperpeak']
    a.f_sd, a.shifts = p['f_sd'], p['shifts']
    a.targetdf, a.mindt, a.targetdt = p['targetdf'], p['mindt'], p['targetdt']


@pytest.mark.parametrize('name', ['noise_s0_10s', 'tonal_s3_20s', 'noise_s8_8s_k8', 'noise_s10_8s_pairgeom'])
def test_find_peaks_landmarks_hashes_methods(name):
    g = load_golden(name)
    a = M.Analyzer()
    _setup(a, g['params'])
    pk = a.find_peaks(g['d'], 11025)
    assert isinstance(pk, list) and isinstance(pk[0], tuple)
    assert np.array_equal(np.array(pk, dtype=np.int32), g['peaks'][0])
    lm = a.peaks2landmarks(pk)
    assert isinstance(lm, list) and np.array_equal(np.array(lm, dtype=np.int32).reshape(-1, 4), g['landmarks0'])
    h = M.landmarks2hashes(lm)
    assert h.dtype == np.int32 and h.shape == (len(lm), 2)
    from oracle import afp_oracle as O
    assert np.array_equal(h, O.landmarks2hashes(g['landmarks0']))
    assert np.array_equal(O.unique_sort_hashes(h), g['hashes'])


@pytest.mark.parametrize('name', ['noise_s1_10s', 'noise_s0_30s_c5', 'noise_s11_8s_sh2', 'hand_silence_then_noise_c5'])
def test_wavfile2hashes_and_stats(tmp_path, name):
    g = load_golden(name)
    fn = str(tmp_path / 'clip.wav')
    _write_wav(fn, g['d'])
    a = M.Analyzer()
    _setup(a, g['params'])
    h = a.wavfile2hashes(fn)
    assert isinstance(h, np.ndarray) and h.dtype == np.int32 and np.array_equal(h, g['hashes'])
    assert a.soundfilecount == 1 and abs(a.soundfiledur - len(g['d']) / 11025.0) < 1e-12
    assert a.soundfiletotaldur == a.soundfiledur
    pk = a.wavfile2peaks(fn, a.shifts)
    if g['params']['shifts'] >= 2:
        assert isinstance(pk, list) and isinstance(pk[0], list) and len(pk) == g['params']['shifts']
        for s, want in enumerate(g['peaks']):
            assert np.array_equal(np.array(pk[s], dtype=np.int32).reshape(-1, 2), want)
    else:
        assert np.array_equal(np.array(pk, dtype=np.int32), g['peaks'][0])
    assert a.soundfilecount == 2


def test_precompute_roundtrip_like_audfprint_py(tmp_path):
    """precompute -> .afpt / .afpk -> wavfile2hashes short-circuits (audfprint.py:70-116)."""
    g = load_golden('noise_s0_10s')
    fn = str(tmp_path / 'track.wav')
    _write_wav(fn, g['d'])
    a = M.Analyzer()
    hashes = a.wavfile2hashes(fn)
    M.hashes_save(str(tmp_path / 'track.afpt'), hashes)
    peaks = a.wavfile2peaks(fn)
    M.peaks_save(str(tmp_path / 'track.afpk'), peaks)
    b = M.Analyzer()
    back = b.wavfile2hashes(str(tmp_path / 'track.afpt'))
    assert isinstance(back, list) and np.array_equal(np.array(back), g['hashes'])
    assert abs(b.soundfiledur - g['hashes'][:, 0].max() * 256 / 11025.0) < 1e-12
    frompk = b.wavfile2hashes(str(tmp_path / 'track.afpk'))          # GPU pairing of a peak file
    assert np.array_equal(np.asarray(frompk), g['hashes'])


def test_ingest_calls_store(tmp_path):
    g = load_golden('noise_s1_10s')
    fn = str(tmp_path / 'x.wav')
    _write_wav(fn, g['d'])

    class Table(object):
        def store(self, name, hashes):
            self.name, self.hashes = name, hashes
    t = Table()
    dur, n = M.Analyzer().ingest(t, fn)
    assert n == len(g['hashes']) and t.name == fn and np.array_equal(t.hashes, g['hashes'])
    assert abs(dur - 10.0) < 1e-9


def test_edge_behaviour(tmp_path, capsys):
    a = M.Analyzer()
    assert a.find_peaks(np.zeros(0, np.float32), 11025) == []
    assert a.find_peaks(np.zeros(11025, np.float32), 11025) == []
    assert 'identically zero' in capsys.readouterr().out          # audfprint_analyze.py:290
    fn = str(tmp_path / 'z.wav')
    _write_wav(fn, np.zeros(5000, np.float32))
    assert a.wavfile2hashes(fn) == []                             # :401-402
    a.shifts = 4
    h = a.wavfile2hashes(fn)
    assert isinstance(h, np.ndarray) and h.shape == (0, 2)
    # unreadable file: IOError when fail_on_error, else "skipping" and empty result (:356-366)
    bad = str(tmp_path / 'missing.wav')
    with pytest.raises(IOError):
        M.Analyzer().wavfile2hashes(bad)
    c = M.Analyzer()
    c.fail_on_error = False
    assert c.wavfile2hashes(bad) == [] and 'skipping' in capsys.readouterr().out


def test_peaks_api_from_peak_lists_multi_unit():
    from audfprint_amd.batch import Extractor
    from oracle import afp_oracle as O
    ex = Extractor.get(0)
    for kw in (dict(), dict(density=70.0, maxpairsperpeak=10, shifts=4), dict(targetdf=10, mindt=1, targetdt=30, maxpairsperpeak=5)):
        prm = O.Params(**kw)
        ex.set_params(**kw)
        clips = [O.synth_noise(70, 3.0), O.synth_tonal(71, 2.0), O.synth_noise(72, 0, nsamp=300)]
        unit_peaks, want_h, want_lm = [], [], []
        for d in clips:
            pls, hs = O.extract(d, prm)
            unit_peaks += pls
            want_h.append(hs)
            want_lm += [O.peaks2landmarks(p, prm) for p in pls]
        res, lms = ex.pairs_from_peaks(unit_peaks, want_hashes=True, want_landmarks=True)
        for i in range(len(clips)):
            assert np.array_equal(res.clip_hashes(i), want_h[i])
        for u in range(len(unit_peaks)):
            assert np.array_equal(lms[u], want_lm[u].astype(np.int32))
