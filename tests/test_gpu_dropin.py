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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.mark.parametrize('name', ['noise_s2_2s_stages', 'tonal_s5_3s_stages'])
def test_semi_private_prune_methods_on_a_given_spectrogram(name):
    """Analyzer._decaying_threshold_fwd_prune / _bwd_prune_peaks take the spectrogram as an ARGUMENT
    (audfprint_analyze.py:199-253): fed the reference's own onset-filtered spectrogram they must return the
    reference's forward mask and, after the backward pass, the mask of the final peak list."""
    g = load_golden(name)
    a = M.Analyzer()
    _setup(a, g['params'])
    sgram = g['sgram']                                            # (256, T) from the live reference
    a_dec = (1 - 0.01 * (a.density * np.sqrt(a.n_hop / 352.8) / 35)) ** (1 / M.OVERSAMP)
    fwd = a._decaying_threshold_fwd_prune(sgram, a_dec)
    want_fwd = np.unpackbits(g['fwd'], axis=0)[:256].astype(np.float64)
    assert fwd.shape == sgram.shape and fwd.dtype == np.float64 and np.array_equal(fwd, want_fwd)
    final = np.zeros(sgram.shape)
    for c, b in g['peaks'][0]:
        final[b, c] = 1
    pk = fwd.copy()
    out = a._decaying_threshold_bwd_prune_peaks(sgram, pk, a_dec)
    assert out is pk and np.array_equal(out, final)               # pruned in place, like the reference
    # an arbitrary mask (every 7th local maximum of each column): compared with the oracle's restatement
    from oracle import afp_oracle as O
    rng = np.random.RandomState(3)
    mask = np.zeros(sgram.shape)
    for t in range(sgram.shape[1]):
        lm = np.nonzero(M.locmax(sgram[:, t]))[0]
        mask[lm[rng.rand(len(lm)) < 0.15], t] = 1
    ref = O.bwd_prune(sgram, mask.copy(), a_dec, O.gauss_table(256, a.f_sd))
    got = a._decaying_threshold_bwd_prune_peaks(sgram, mask.copy(), a_dec)
    assert np.array_equal(got, ref) and np.all(got <= mask)


def test_float64_waveform_is_not_rounded_to_float32():
    """find_peaks(d) works in the dtype of d (stft.py:87-93 keeps float64); a float64 waveform whose
    samples are not float32-representable must go through the float64 ingest and match the oracle on float64."""
    from oracle import afp_oracle as O
    rng = np.random.RandomState(21)
    d64 = rng.randn(5 * 11025) * 0.1 + 1e-9 * rng.randn(5 * 11025)       # not representable in float32
    assert not np.array_equal(d64.astype(np.float32).astype(np.float64), d64)
    a = M.Analyzer()
    pk = a.find_peaks(d64, 11025)
    ref = O.find_peaks(d64, O.Params())
    assert np.array_equal(np.array(pk, dtype=np.int32).reshape(-1, 2), ref)
    # int16 array handed to find_peaks: the reference does NOT divide by 32768 (that is audio_read's job)
    i16 = np.round(np.clip(d64, -1, 1) * 32767).astype(np.int16)
    pk16 = a.find_peaks(i16, 11025)
    assert np.array_equal(np.array(pk16, dtype=np.int32).reshape(-1, 2), O.find_peaks(i16.astype(np.float64), O.Params()))


def test_float64_waveform_of_extreme_scale():
    """|S|^2 must not under/overflow where the reference's np.abs does not; a float64 waveform scaled by
    2^-700 / 2^+700 gives the reference the same peaks as the unscaled one (checked on the live reference), and us too."""
    from oracle import afp_oracle as O
    rng = np.random.RandomState(22)
    d64 = rng.randn(4 * 11025) * 0.1
    ref = O.find_peaks(d64, O.Params())
    a = M.Analyzer()
    for k in (-700, -160, 600, 900):
        dk = np.ldexp(d64, k)
        assert np.array_equal(O.find_peaks(dk, O.Params()), ref)
        pk = a.find_peaks(dk, 11025)
        assert np.array_equal(np.array(pk, dtype=np.int32).reshape(-1, 2), ref), k
    # float32 denormals (1e-40) need no rescaling: their squares are far inside the float64 range
    d32 = (rng.randn(4 * 11025) * 1e-40).astype(np.float32)
    pk = a.find_peaks(d32, 11025)
    assert np.array_equal(np.array(pk, dtype=np.int32).reshape(-1, 2), O.find_peaks(d32.astype(np.float64), O.Params()))


def test_lone_click_warns_tie_prone():
    import warnings
    d = np.zeros(3 * 11025, np.float32)
    d[11025] = 0.5
    a = M.Analyzer()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        a.find_peaks(d, 11025)
    assert any('rounding noise' in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        a.find_peaks(load_golden('noise_s1_10s')['d'], 11025)
    assert not w


def _child_extract(name, q):
    """Runs in a child process: the pickled Analyzer creates its own device context there."""
    try:
        import pickle
        g = load_golden(name)
        a = pickle.loads(q['an'])
        pk = a.find_peaks(g['d'], 11025)
        q['out'].put((name, np.array(pk, dtype=np.int32).reshape(-1, 2)))
    except Exception as e:  # pragma: no cover
        q['out'].put((name, repr(e)))


@pytest.mark.parametrize('method', ['spawn', 'fork'])
def test_analyzer_in_child_processes(method):
    """The process model the boundary promises (audfprint.py:217-224, 249-251): the Analyzer is pickled into
    joblib workers / inherited by forked multiprocessing children, each of which builds its OWN device context.
    `fork` children are started from a clean helper process that has not touched the GPU (as audfprint.py forks
    before any analysis runs): a HIP context does not survive fork()."""
    import multiprocessing as mp
    import pickle
    import subprocess
    names = ['noise_s0_10s', 'tonal_s3_20s']
    if method == 'fork':
        # this pytest process already holds a HIP context -> do the forking in a fresh interpreter
        code = (
            "import sys, pickle, numpy as np, multiprocessing as mp\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import audfprint_amd.audfprint_analyze as M\n"
            "from conftest import load_golden\n"
            "def work(name, q):\n"
            "    g = load_golden(name); a = AN\n"
            "    q.put((name, np.array(a.find_peaks(g['d'], 11025), dtype=np.int32).reshape(-1, 2).tolist()))\n"
            "AN = M.Analyzer()\n"
            "if __name__ == '__main__':\n"
            "    ctx = mp.get_context('fork'); q = ctx.Queue()\n"
            "    ps = [ctx.Process(target=work, args=(n, q)) for n in %r]\n"
            "    [p.start() for p in ps]\n"
            "    res = dict(q.get(timeout=300) for _ in ps)\n"
            "    [p.join() for p in ps]\n"
            "    # the parent uses the GPU only AFTER its children were forked, then forks again: the grandchildren must\n"
            "    # not destroy the parent's context (Extractor.close is a no-op outside the creating process)\n"
            "    g = load_golden(%r); par = np.array(AN.find_peaks(g['d'], 11025), dtype=np.int32).reshape(-1, 2).tolist()\n"
            "    import pickle, sys\n"
            "    sys.stdout.buffer.write(pickle.dumps((res, par)))\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), names, names[0])
        out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        res, par = pickle.loads(out.stdout[out.stdout.index(b'\x80'):])
        for n in names:
            assert np.array_equal(np.array(res[n], np.int32).reshape(-1, 2), load_golden(n)['peaks'][0]), n
        assert np.array_equal(np.array(par, np.int32).reshape(-1, 2), load_golden(names[0])['peaks'][0])
        return
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    an = pickle.dumps(M.Analyzer())
    ps = [ctx.Process(target=_child_extract, args=(n, dict(an=an, out=q))) for n in names]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join()
    for n in names:
        assert not isinstance(res[n], str), res[n]
        assert np.array_equal(res[n], load_golden(n)['peaks'][0]), n


def test_cli_call_order_precompute_then_new_on_the_gpu(tmp_path):
    """BASELINE configs[0] plumbing END TO END on the real library: the call sequence of the reference CLI --
    setup_analyzer's attribute writes (audfprint.py:280-299), then per file `analyzer.wavfile2hashes` +
    `audfprint_analyze.hashes_save` (precompute, :70-116), then `analyzer.ingest(hash_tab, file)` per file and the
    "Added N hashes" report (new, :173-186), then `new` again from the precomputed .afpt files.  When the
    reference tree is mounted the REAL audfprint.do_cmd / file_precompute drive it (unchanged); on the GPU box,
    which has no reference tree, a stand-in issues the very same calls in the same order."""
    from oracle import afp_oracle as O
    names, clips = [], []
    for i in range(3):
        d = O.synth_noise(9700 + i, 4.0 + i)
        fn = str(tmp_path / ('src%d.wav' % i))
        _write_wav(fn, d)
        names.append(fn)
        clips.append(d)
    want = [O.extract(d, O.Params(density=35.0, maxpairsperpeak=5))[1] for d in clips]
    outdir = str(tmp_path / 'pre')
    # the reference tree: only where the caller names one (AFP_REF_DIR).  The GPU box has none -- the reference's sources do
    # not travel -- and this test never looks for one on its own (rounds 3-5 shipped a scratch copy for one evidence run each:
    # profiles/r03_real_cli_on_gpu.log, r05_real_cli_on_gpu.log; that mechanism is retired)
    ref = next((p for p in (os.environ.get('AFP_REF_DIR'),)
                if p and os.path.isfile(os.path.join(p, 'audfprint.py'))), '/nonexistent')
    reports = []
    print('reference CLI tree: %s' % (ref if os.path.isdir(ref) else 'absent (stand-in issues the same calls)'))
    if os.path.isdir(ref):                                           # the real CLI module, unchanged
        sys.modules['audfprint_analyze'] = M
        docopt = types.ModuleType('docopt')
        docopt.docopt = lambda *a, **k: {}
        sys.modules['docopt'] = docopt
        sys.path.insert(0, ref)
        try:
            import audfprint
            import hash_table
            an = audfprint.setup_analyzer({'--density': '35', '--pks-per-frame': '5', '--fanout': '5', '--freq-sd': '30.0',
                                           '--shifts': '0', '--samplerate': '11025', '--continue-on-error': False, 'match': False})
            audfprint.do_cmd('precompute', an, None, iter(names), None, outdir, 'hashes', reports.extend)
            ht = hash_table.HashTable(hashbits=20, depth=100, maxtime=16384)
            audfprint.do_cmd('new', an, ht, iter(names), None, None, None, reports.extend)
        finally:
            sys.path.remove(ref)
            for m in ('audfprint', 'audfprint_match', 'hash_table', 'audfprint_analyze', 'docopt', 'stft'):
                sys.modules.pop(m, None)
    else:                                                            # the same calls, in the same order
        an = M.Analyzer()
        an.density, an.maxpksperframe, an.maxpairsperpeak, an.f_sd = 35.0, 5, 5, 30.0      # audfprint.py:285-291
        an.shifts, an.target_sr, an.n_fft, an.n_hop, an.fail_on_error = 1, 11025, 512, 256, True
        for fn in names:                                             # file_precompute_peaks_or_hashes, :70-116
            relname = '/'.join(c for c in fn.split('/') if c not in ('.', '..', ''))
            opf = os.path.join(outdir, os.path.splitext(relname)[0] + M.PRECOMPEXT)
            output = an.wavfile2hashes(fn)
            assert len(output) != 0
            os.makedirs(os.path.split(opf)[0], exist_ok=True)
            M.hashes_save(opf, output)
            reports.append("wrote " + opf + " ( %d %s, %.3f sec)" % (len(output), 'hashes', an.soundfiledur))
        ht = O.OracleHashTable(hashbits=20, depth=100)
        ht.store = lambda name, h, _s=ht.store: _s(name, h, None)    # (store() never overflows here: no RNG draw)
        tothashes = 0
        for fn in names:                                             # do_cmd 'new', :173-186
            dur, nhash = an.ingest(ht, fn)
            tothashes += nhash
        reports.append("Added " + str(tothashes) + " hashes (%.1f hashes/sec)" % (tothashes / float(an.soundfiletotaldur)))
    # what came out: the .afpt bytes, the table, the accounting
    for fn, h in zip(names, want):
        relname = '/'.join(c for c in fn.split('/') if c not in ('.', '..', ''))
        opf = os.path.join(outdir, os.path.splitext(relname)[0] + '.afpt')
        assert open(opf, 'rb').read() == b'audfprinthashV00' + h.astype('<i4').tobytes()
        assert any(('%d hashes' % len(h)) in r for r in reports)
    assert list(ht.names) == names and int(ht.counts.sum()) == sum(len(h) for h in want)
    assert [int(x) for x in ht.hashesperid] == [len(h) for h in want]
    assert an.soundfilecount == 6 and abs(an.soundfiletotaldur - 2 * (4 + 5 + 6)) < 1e-6
    assert any(r.startswith('Added %d hashes' % sum(len(h) for h in want)) for r in reports)
    # `new` from the precomputed files: the .afpt short-circuit of wavfile2hashes (audfprint_analyze.py:391-398)
    ht2 = O.OracleHashTable(hashbits=20, depth=100)
    ht2.store = lambda name, h, _s=ht2.store: _s(name, h, None)
    b = M.Analyzer()
    for fn in names:
        relname = '/'.join(c for c in fn.split('/') if c not in ('.', '..', ''))
        b.ingest(ht2, os.path.join(outdir, os.path.splitext(relname)[0] + '.afpt'))
    assert np.array_equal(ht2.counts, np.asarray(ht.counts)) and np.array_equal(ht2.table, np.asarray(ht.table))


def test_ncores_2_on_one_gpu():
    """`--ncores 2` through the drop-in on a one-GPU box -- forked children (new) and joblib workers
    (precompute) each open their own context on GPU (ordinal - 1) mod 1 = 0, and what they produce equals what one process
    produces, file by file (tests/_ncores_gpu_run.py, in a fresh interpreter: the parent must not hold a HIP context when it
    forks).  With AFP_REF_DIR the reference's own multiproc_add / do_cmd_multiproc drive it."""
    import subprocess
    env = dict(os.environ)
    env.pop('AFP_DEVICE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_ncores_gpu_run.py')], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=900, text=True)
    print(out.stdout)
    assert out.returncode == 0 and 'NCORES2 OK' in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_bulk_forms_equal_a_loop_over_the_per_file_methods(tmp_path, capsys):
    """Analyzer.wavfiles2hashes / ingest_many (round 6: the batch API behind the Analyzer's own parameters): the same list a
    loop over wavfile2hashes / ingest gives, element for element -- arrays, [] for a file without peaks, the python list of an
    .afpt file, an .afpk file -- the same bookkeeping, the same "identically zero" warning, for one shift and for four."""
    from oracle import afp_oracle as O
    files = []
    for i, d in enumerate([O.synth_noise(7100, 6.0), O.synth_tonal(7101, 4.0), np.zeros(9000, np.float32),
                           O.synth_noise(7102, 0, nsamp=200), O.synth_noise(7103, 11.0)]):
        fn = str(tmp_path / ('bulk%d.wav' % i))
        _write_wav(fn, d)
        files.append(fn)
    pre = M.Analyzer()
    M.hashes_save(str(tmp_path / 'pre.afpt'), pre.wavfile2hashes(files[0]))
    M.peaks_save(str(tmp_path / 'pre.afpk'), pre.wavfile2peaks(files[1]))
    files = files[:2] + [str(tmp_path / 'pre.afpt')] + files[2:] + [str(tmp_path / 'pre.afpk')]
    for shifts in (1, 4):
        a, b = M.Analyzer(), M.Analyzer()
        a.shifts = b.shifts = shifts
        capsys.readouterr()
        want = [a.wavfile2hashes(f) for f in files]
        warn_a = capsys.readouterr().out.count('identically zero')
        got = b.wavfiles2hashes(files)
        warn_b = capsys.readouterr().out.count('identically zero')
        assert warn_a == warn_b >= 1
        assert len(got) == len(want)
        for f, w, g in zip(files, want, got):
            assert type(w) is type(g), (f, type(w), type(g))
            assert np.array_equal(np.asarray(w), np.asarray(g)), f
        assert (a.soundfilecount, a.soundfiledur) == (b.soundfilecount, b.soundfiledur)
        assert abs(a.soundfiletotaldur - b.soundfiletotaldur) < 1e-9
    # ingest_many == a loop over ingest: same table, same (dur, nhashes) list
    ta, tb_ = O.OracleHashTable(hashbits=20, depth=100), O.OracleHashTable(hashbits=20, depth=100)
    ta.store = lambda name, h, _s=ta.store: _s(name, np.asarray(h).reshape(-1, 2), None)
    tb_.store = lambda name, h, _s=tb_.store: _s(name, np.asarray(h).reshape(-1, 2), None)
    a, b = M.Analyzer(), M.Analyzer()
    ra = [a.ingest(ta, f) for f in files]
    rb = b.ingest_many(tb_, files)
    assert ra == rb
    assert ta.names == tb_.names and np.array_equal(ta.table, tb_.table) and np.array_equal(ta.counts, tb_.counts)
