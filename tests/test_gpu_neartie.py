"""the near-tie guard of the threshold passes (afp_set_neartie_eps, AFP_UNIT_NEARTIE; off unless asked for:
it costs 3 % of a C3 step).  At the epsilon bench.py's parity passes use (1e-11: a hundred times the library's
log-spectrogram difference from numpy's) nothing in the fixture set or in random noise is marked -- every decision stands
by a margin; with the epsilon forced up to 1e-3 the marks fire, compact-path batches are re-run on the dense path and
counted, and the integers still equal the oracle's."""
import numpy as np
import pytest

from conftest import SPARSE_FRAME, golden_names, load_golden

pytestmark = pytest.mark.gpu
PKEYS = ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts', 'targetdf', 'mindt', 'targetdt')
PATHS = {'dense': dict(compact=0, seg=0), 'compact': dict(compact=1, seg=0), 'segments': dict(compact=0, seg=1, seg_len=16, seg_warm=32)}


@pytest.fixture(scope='module')
def ex():
    from audfprint_amd.batch import Extractor
    e = Extractor.get(0)
    yield e
    e.set_pipeline()
    e.set_neartie_eps(0.0)            # the library's default: off


@pytest.mark.parametrize('path', sorted(PATHS))
def test_default_epsilon_marks_no_fixture_and_no_noise(ex, path):
    from audfprint_amd import _lib
    from oracle import afp_oracle as O
    ex.set_neartie_eps(1e-11)
    ex.set_pipeline(**PATHS[path])
    marked = []
    for name in golden_names():
        if name in SPARSE_FRAME:
            continue
        g = load_golden(name)
        ex.set_params(**{k: g['params'][k] for k in PKEYS})
        r = ex.extract(clips=[g['d']], want_hashes=True, want_peaks=True)
        if np.any(r.unit_flags & _lib.UNIT_NEARTIE):
            marked.append(name)
        assert np.array_equal(r.clip_hashes(0), g['hashes']), (path, name)
    assert not marked, (path, marked)
    ex.set_params()
    clips = [O.synth_noise(500 + i, 30.0) for i in range(96)]
    r = ex.extract(clips=clips, want_hashes=True, want_peaks=False)
    st = ex.path_stats()
    assert st['near_tie_units'] == 0 and not st['near_tie_redone'] and not np.any(r.unit_flags & _lib.UNIT_NEARTIE), (path, st)


@pytest.mark.parametrize('path', sorted(PATHS))
def test_forced_epsilon_fires_redoes_the_compact_batch_and_keeps_the_integers(ex, path):
    from audfprint_amd import _lib
    from oracle import afp_oracle as O
    ex.set_params()
    clips = [O.synth_noise(700 + i, 20.0) for i in range(24)] + [O.synth_tonal(731, 10.0)]
    prm = O.Params()
    want = [O.extract(d, prm) for d in clips]
    ex.set_pipeline(**PATHS[path])
    try:
        ex.set_neartie_eps(1e-3)
        r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        st = ex.path_stats()
        nmarked = int(np.count_nonzero(r.unit_flags & _lib.UNIT_NEARTIE))
        # 20 s of noise holds ~1700 decisive comparisons within a few units of each other: every clip meets one closer than 1e-3
        assert nmarked >= len(clips) // 2 and st['near_tie_units'] == nmarked, (path, nmarked, st)
        assert st['near_tie_redone'] == (path == 'compact'), (path, st)
        if path == 'compact':
            assert not st['compact'] and st['near_tie_redone_total'] >= 1, st        # the results are the dense path's
        for i, (pls, hs) in enumerate(want):
            assert np.array_equal(r.unit_peaks(i, 0), pls[0]) and np.array_equal(r.clip_hashes(i), hs), (path, i)
        # epsilon 0: the guard is off, nothing is marked, nothing re-run
        ex.set_neartie_eps(0.0)
        r0 = ex.extract(clips=clips, want_hashes=True, want_peaks=False)
        st0 = ex.path_stats()
        assert st0['near_tie_units'] == 0 and not st0['near_tie_redone'] and not np.any(r0.unit_flags & _lib.UNIT_NEARTIE)
        assert st0['compact'] == (path == 'compact')
        assert np.array_equal(r0.hashes, r.hashes)
    finally:
        ex.set_neartie_eps(0.0)
        ex.set_pipeline()


def test_bad_epsilon_is_refused(ex):
    from audfprint_amd import _lib
    with pytest.raises(_lib.AfpError):
        ex.set_neartie_eps(-1.0)
    with pytest.raises(_lib.AfpError):
        ex.set_neartie_eps(float('nan'))


@pytest.mark.parametrize('cfg', [dict(), dict(density=70.0, maxpairsperpeak=10, shifts=4)], ids=['c3', 'c5'])
def test_unguarded_and_guarded_instantiations_give_the_same_rows_on_a_compact_batch(ex, cfg):
    """the kernels bench.py TIMES are the GUARD=false instantiations (k_scan_small<..., false>), the near-tie
    pass a separately compiled sibling (GUARD=true).  A compact-path batch of >= 1024 units (what C3 / C5 run), 64 distinct
    clips: both give identical rows, and both equal the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import afp_oracle as O
    shifts = cfg.get('shifts', 1)
    ndist, nclips, secs = 64, 1024 // shifts + 32, 12.0
    pool = [O.synth_noise(8800 + i, secs) for i in range(ndist)]
    clips = [pool[i % ndist] for i in range(nclips)]
    prm = O.Params(**cfg)
    with ThreadPoolExecutor(8) as tp:
        want = list(tp.map(lambda d: O.extract(d, prm)[1], pool))
    ex.set_params(**cfg)
    ex.set_pipeline()                          # the library's own rule: >= 768 units -> compact
    res = {}
    try:
        for name, eps in (('unguarded', 0.0), ('guarded', 1e-11)):
            ex.set_neartie_eps(eps)
            r = ex.extract(clips=clips, want_hashes=True, want_peaks=False)
            st = ex.path_stats()
            assert st['compact'] and not st['redone_dense'] and not st['near_tie_redone'], (name, st)
            assert st['near_tie_units'] == 0, (name, st)
            res[name] = r
    finally:
        ex.set_neartie_eps(0.0)
    a, b = res['unguarded'], res['guarded']
    assert np.array_equal(a.hash_offsets, b.hash_offsets) and np.array_equal(a.hashes, b.hashes)
    for name, r in res.items():
        for i in range(nclips):
            assert np.array_equal(r.clip_hashes(i), want[i % ndist]), (name, i)
