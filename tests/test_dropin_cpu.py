"""CPU: the drop-in module's surface, file formats and pickling (no GPU compute)."""
import os
import pickle
import sys

import numpy as np
import pytest

import audfprint_amd.audfprint_analyze as M

REF = '/root/reference'

REF_NAMES = ['PRECOMPEXT', 'PRECOMPPKEXT', 'locmax', 'DENSITY', 'OVERSAMP', 'N_FFT', 'N_HOP', 'HPF_POLE',
             'F1_BITS', 'DF_BITS', 'DT_BITS', 'B1_MASK', 'B1_SHIFT', 'DF_MASK', 'DF_SHIFT', 'DT_MASK',
             'landmarks2hashes', 'hashes2landmarks', 'Analyzer', 'HASH_FMT', 'HASH_MAGIC', 'PEAK_FMT',
             'PEAK_MAGIC', 'hashes_save', 'hashes_load', 'peaks_save', 'peaks_load', 'extract_features',
             'glob2hashtable', 'g2h_analyzer']       # (the reference's ad-hoc `local_tester` is not interface)
METHODS = ['find_peaks', 'peaks2landmarks', 'wavfile2peaks', 'wavfile2hashes', 'ingest', 'spreadpeaks',
           'spreadpeaksinvector', '_decaying_threshold_fwd_prune', '_decaying_threshold_bwd_prune_peaks']
ATTRS = dict(density=20.0, target_sr=11025, n_fft=512, n_hop=256, shifts=1, f_sd=30.0, maxpksperframe=5,
             maxpairsperpeak=3, targetdf=31, mindt=2, targetdt=63, soundfiledur=0.0, soundfiletotaldur=0.0,
             soundfilecount=0, fail_on_error=True)


def test_module_surface():
    for n in REF_NAMES:
        assert hasattr(M, n), n
    a = M.Analyzer()
    for k, v in ATTRS.items():
        assert getattr(a, k) == v, k
    for m in METHODS:
        assert callable(getattr(a, m))
    assert M.Analyzer(70.0).density == 70.0


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')
def test_surface_and_constants_equal_the_reference_module():
    sys.path.insert(0, REF)
    try:
        import audfprint_analyze as R
        for n in REF_NAMES:
            assert hasattr(R, n), n
            if isinstance(getattr(R, n), (int, float, str, bytes)):
                assert getattr(R, n) == getattr(M, n), n
        ra, ma = R.Analyzer(), M.Analyzer()
        assert {k: v for k, v in vars(ra).items()} == {k: v for k, v in vars(ma).items()}
        v = np.random.RandomState(0).randn(300)
        v[10:14] = v[10]
        assert np.array_equal(R.locmax(v), M.locmax(v))
        assert np.array_equal(R.locmax(v, indices=True), M.locmax(v, indices=True))
        h = [(3, 0xABCDE), (9, 0x12345), (11, 0xFFFFF)]
        assert R.hashes2landmarks(h) == M.hashes2landmarks(h)
        # the vector helpers (audfprint_analyze.py:153-197): same values, bit for bit
        for width in (30.0, 4.0):
            assert np.array_equal(ra.spreadpeaksinvector(v[:256], width), ma.spreadpeaksinvector(v[:256], width))
        pk = [(5, 1.5), (200, 0.25), (255, 2.0)]
        assert np.array_equal(ra.spreadpeaks(pk, npoints=256, width=30.0), ma.spreadpeaks(pk, npoints=256, width=30.0))
        base = np.abs(v[:100])
        assert np.array_equal(ra.spreadpeaks(pk[:1], width=7.0, base=base), ma.spreadpeaks(pk[:1], width=7.0, base=base))
        assert {k: v_ for k, v_ in vars(ra).items() if not k.startswith('_Analyzer__')} == vars(ma)
    finally:
        sys.path.remove(REF)
        for m in ('audfprint_analyze', 'stft', 'audio_read', 'hash_table'):
            sys.modules.pop(m, None)


def test_analyzer_pickles_and_holds_no_device_state():
    a = M.Analyzer(35.0)
    a.shifts = 4
    a.soundfiletotaldur = 12.5
    b = pickle.loads(pickle.dumps(a))
    assert vars(a) == vars(b)
    for v in vars(a).values():
        assert isinstance(v, (int, float, bool))


def test_file_formats_roundtrip_and_bytes(tmp_path):
    rng = np.random.RandomState(1)
    hashes = np.stack([np.sort(rng.randint(0, 5000, 300)), rng.randint(0, 1 << 20, 300)], axis=1).astype(np.int32)
    fn = str(tmp_path / 'x.afpt')
    M.hashes_save(fn, hashes)
    raw = open(fn, 'rb').read()
    assert raw[:16] == b'audfprinthashV00' and len(raw) == 16 + 8 * 300
    assert raw[16:] == hashes.astype('<i4').tobytes()
    back = M.hashes_load(fn)
    assert isinstance(back, list) and isinstance(back[0], tuple) and np.array_equal(np.array(back), hashes)
    peaks = [(1, 5), (1, 200), (7, 33)]
    fp = str(tmp_path / 'x.afpk')
    M.peaks_save(fp, peaks)
    assert open(fp, 'rb').read()[:16] == b'audfprintpeakV00' and M.peaks_load(fp) == peaks
    M.hashes_save(fn, [])
    assert open(fn, 'rb').read() == b'audfprinthashV00' and M.hashes_load(fn) == []
    with pytest.raises(IOError):
        M.peaks_load(fn)            # wrong magic


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')
def test_file_bytes_equal_reference_writer(tmp_path):
    sys.path.insert(0, REF)
    try:
        import audfprint_analyze as R
        rng = np.random.RandomState(2)
        hashes = np.stack([np.sort(rng.randint(0, 900, 77)), rng.randint(0, 1 << 20, 77)], axis=1).astype(np.int32)
        a, b = str(tmp_path / 'a.afpt'), str(tmp_path / 'b.afpt')
        R.hashes_save(a, hashes)
        M.hashes_save(b, hashes)
        assert open(a, 'rb').read() == open(b, 'rb').read()
        assert R.hashes_load(b) == M.hashes_load(a)
        pk = [(0, 3), (0, 250), (5, 77)]
        R.peaks_save(a, pk)
        M.peaks_save(b, pk)
        assert open(a, 'rb').read() == open(b, 'rb').read() and R.peaks_load(b) == M.peaks_load(a)
    finally:
        sys.path.remove(REF)
        for m in ('audfprint_analyze', 'stft', 'audio_read', 'hash_table'):
            sys.modules.pop(m, None)


def test_empty_input_shortcuts_need_no_gpu():
    a = M.Analyzer()
    assert a.find_peaks([], 11025) == []                       # audfprint_analyze.py:273-274
    assert a.peaks2landmarks([]) == []
    assert M.landmarks2hashes([]).shape == (0, 2) and M.landmarks2hashes([]).dtype == np.int32


class _FakeExtractor(object):
    """Stands in for the device context: records what the Analyzer asks for and hands back crafted results."""

    def __init__(self, hashes, peaks):
        self.hashes, self.peaks, self.calls = hashes, peaks, []

    def extract(self, clips=None, want_hashes=True, want_peaks=False, **kw):
        from audfprint_amd.batch import BatchResult
        self.calls.append((len(clips), bool(want_hashes), bool(want_peaks)))
        r = BatchResult()
        r.nclips, r.shifts = 1, 1
        if want_hashes:
            r.hashes = np.asarray(self.hashes, np.int32).reshape(-1, 2)
            r.hash_offsets = np.array([0, len(r.hashes)], np.int64)
        if want_peaks:
            r.peaks = np.asarray(self.peaks, np.int32).reshape(-1, 2)
            r.peak_offsets = np.array([0, len(r.peaks)], np.int64)
        r.unit_flags = np.zeros(1, np.int32)
        return r


@pytest.mark.parametrize('hashes,peaks,expect', [
    ([(1, 77), (2, 88)], [(1, 5)], 'rows'),          # the common case: rows back, ONE extraction, no peak list asked for
    ([], [], 'list'),                                # no peak at all: the reference returns [] (audfprint_analyze.py:401-402)
    ([], [(4, 9)], 'empty_rows'),                    # peaks that pair into nothing: an empty (0,2) array through :404-422
])
def test_wavfile2hashes_asks_for_the_peak_list_only_when_there_are_no_hashes(monkeypatch, hashes, peaks, expect):
    a = M.Analyzer()
    fake = _FakeExtractor(hashes, peaks)
    monkeypatch.setattr(a, '_extractor', lambda shifts: fake)
    monkeypatch.setattr(a, '_read_audio', lambda fn: (np.ones(2000, np.float32), 11025))
    out = a.wavfile2hashes('x.wav')
    if expect == 'rows':
        assert np.array_equal(out, np.asarray(hashes, np.int32)) and fake.calls == [(1, True, False)]
    elif expect == 'list':
        assert isinstance(out, list) and out == [] and fake.calls == [(1, True, False), (1, False, True)]
    else:
        assert isinstance(out, np.ndarray) and out.shape == (0, 2) and fake.calls == [(1, True, False), (1, False, True)]
    assert a.soundfilecount == 1 and abs(a.soundfiledur - 2000 / 11025) < 1e-12
    # several shifts: never a peak list, an array even when empty (the concatenate / unique path)
    a.shifts = 4
    fake2 = _FakeExtractor([], [])
    monkeypatch.setattr(a, '_extractor', lambda shifts: fake2)
    out = a.wavfile2hashes('x.wav')
    assert isinstance(out, np.ndarray) and out.shape == (0, 2) and fake2.calls == [(1, True, False)]


class _FakeBatchExtractor(object):
    """A device stand-in for several clips per call: clip k of a call yields the rows / peaks filed under its FIRST SAMPLE."""

    def __init__(self, table):
        self.table, self.calls = table, []

    def extract(self, clips=None, want_hashes=True, want_peaks=False, **kw):
        from audfprint_amd.batch import BatchResult
        self.calls.append((len(clips), bool(want_hashes), bool(want_peaks)))
        r = BatchResult()
        r.nclips, r.shifts = len(clips), 1
        keys = [int(round(float(c[0]) * 100)) for c in clips]
        if want_hashes:
            rows = [np.asarray(self.table[k][0], np.int32).reshape(-1, 2) for k in keys]
            r.hashes = np.concatenate(rows) if rows else np.zeros((0, 2), np.int32)
            r.hash_offsets = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
        if want_peaks:
            pk = [np.asarray(self.table[k][1], np.int32).reshape(-1, 2) for k in keys]
            r.peaks = np.concatenate(pk) if pk else np.zeros((0, 2), np.int32)
            r.peak_offsets = np.concatenate([[0], np.cumsum([len(x) for x in pk])]).astype(np.int64)
        r.unit_flags = np.zeros(len(clips), np.int32)
        return r


def test_bulk_forms_keep_the_order_types_and_bookkeeping_of_a_loop(monkeypatch, tmp_path):
    """Analyzer.wavfiles2hashes / ingest_many (round 6) without a GPU: ONE extraction for all audio files, a second one only
    for the clips that came back without rows, precomputed files passed through in list order, the bookkeeping of a loop."""
    table = {1: ([(1, 11), (2, 22)], [(1, 5)]),      # rows
             2: ([], []),                            # no peak at all -> []
             3: ([], [(4, 9)]),                      # peaks that pair into nothing -> empty (0, 2) array
             4: ([(7, 70)], [(7, 1)])}
    audio = {'a.wav': np.full(1100, 0.01, np.float32), 'b.wav': np.full(2200, 0.02, np.float32), 'c.wav': np.full(3300, 0.03, np.float32),
             'd.wav': np.full(4400, 0.04, np.float32), 'e.wav': np.zeros(0, np.float32)}
    pre = str(tmp_path / 'p.afpt')
    M.hashes_save(pre, np.array([[3, 33], [9, 99]], np.int32))
    files = ['a.wav', 'b.wav', pre, 'c.wav', 'e.wav', 'd.wav']

    def make():
        a = M.Analyzer()
        fake = _FakeBatchExtractor(table)
        monkeypatch.setattr(a, '_extractor', lambda shifts: fake)
        monkeypatch.setattr(a, '_read_audio', lambda fn: (audio[fn], 11025))
        return a, fake
    a, fa = make()
    want = [a.wavfile2hashes(f) for f in files]
    b, fb = make()
    got = b.wavfiles2hashes(files)
    assert fb.calls == [(4, True, False), (2, False, True)]            # a, b, c, d in one launch; b and c asked for peaks
    assert len(fa.calls) == 4 + 2
    for w, g in zip(want, got):
        assert type(w) is type(g) and np.array_equal(np.asarray(w), np.asarray(g))
    assert isinstance(got[1], list) and got[1] == [] and isinstance(got[3], np.ndarray) and got[3].shape == (0, 2)
    assert isinstance(got[2], list) and got[2] == [(3, 33), (9, 99)] and got[4] == []
    assert (a.soundfilecount, a.soundfiledur) == (b.soundfilecount, b.soundfiledur) == (6, 4400 / 11025)
    assert abs(a.soundfiletotaldur - b.soundfiletotaldur) < 1e-12

    class _HT(object):
        def __init__(self):
            self.rows = []

        def store(self, name, h):
            self.rows.append((name, np.asarray(h).reshape(-1, 2).tolist()))
    a, _ = make()
    b, _ = make()
    ta, tb = _HT(), _HT()
    assert [a.ingest(ta, f) for f in files] == b.ingest_many(tb, files)
    assert ta.rows == tb.rows and [n for n, _ in tb.rows] == files
    # four shifts: never a peak list, empty arrays stay arrays
    b, fb = make()
    b.shifts = 4
    got = b.wavfiles2hashes(['b.wav', 'e.wav'])
    assert fb.calls == [(1, True, False)] and all(isinstance(g, np.ndarray) and g.shape == (0, 2) for g in got)
