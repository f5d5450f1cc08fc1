"""Integer stages (peaks2landmarks, landmarks2hashes, unique/sort) on random peak lists:
CPU: oracle vs the live reference (hypothesis); GPU: afp_pairs_from_peaks vs the oracle."""
import os
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import afp_oracle as O

REF = '/root/reference'


def _random_peaks(rng, nframes, density, maxk):
    rows = []
    for t in range(nframes):
        k = min(maxk, rng.poisson(density))
        for b in sorted(rng.choice(256, size=k, replace=False)):
            rows.append((t, int(b)))
    return np.array(rows, dtype=np.int32).reshape(-1, 2)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')
@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 10 ** 6), nframes=st.integers(1, 160), density=st.floats(0.05, 3.0),
       fanout=st.integers(1, 12), targetdf=st.integers(1, 40), mindt=st.integers(0, 4), targetdt=st.integers(1, 70))
def test_oracle_pairing_equals_reference(seed, nframes, density, fanout, targetdf, mindt, targetdt):
    sys.path.insert(0, REF)
    try:
        import audfprint_analyze as R
    finally:
        sys.path.remove(REF)
    rng = np.random.RandomState(seed)
    pk = _random_peaks(rng, nframes, density, 5)
    an = R.Analyzer()
    an.maxpairsperpeak, an.targetdf, an.mindt, an.targetdt = fanout, targetdf, mindt, targetdt
    prm = O.Params(maxpairsperpeak=fanout, targetdf=targetdf, mindt=mindt, targetdt=targetdt)
    ref_lm = an.peaks2landmarks([(int(c), int(b)) for c, b in pk])
    got_lm = O.peaks2landmarks(pk, prm)
    assert np.array_equal(np.array(ref_lm, dtype=np.int64).reshape(-1, 4), got_lm)
    assert np.array_equal(R.landmarks2hashes(ref_lm), O.landmarks2hashes(got_lm))


@pytest.fixture(scope='module', params=['fused', 'generic'])
def pair_ex(request):
    """An Extractor using k_pairmerge (default) or the generic k_pair + k_merge kernels."""
    from audfprint_amd.batch import Extractor
    if request.param == 'generic':
        os.environ['AFP_GENERIC_PAIR'] = '1'
    try:
        e = Extractor(0)
    finally:
        os.environ.pop('AFP_GENERIC_PAIR', None)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(6))
def test_gpu_pairing_of_random_peak_lists(pair_ex, seed):
    ex = pair_ex
    rng = np.random.RandomState(100 + seed)
    kw = dict(maxpairsperpeak=int(rng.choice([1, 3, 10, 25])), targetdf=int(rng.choice([5, 31, 40])),
              mindt=int(rng.choice([0, 1, 2])), targetdt=int(rng.choice([8, 63, 100])),
              maxpksperframe=int(rng.choice([3, 5, 9])), shifts=int(rng.choice([1, 1, 2, 4])))
    prm = O.Params(**kw)
    ex.set_params(**kw)
    nclips = 5
    unit_peaks = [_random_peaks(rng, int(rng.randint(0, 400)), float(rng.uniform(0.1, 2.5)), kw['maxpksperframe'])
                  for _ in range(nclips * prm.shifts)]
    res, lms = ex.pairs_from_peaks(unit_peaks, want_hashes=True, want_landmarks=True)
    for c in range(nclips):
        hs = [O.landmarks2hashes(O.peaks2landmarks(unit_peaks[c * prm.shifts + s], prm)) for s in range(prm.shifts)]
        allh = np.concatenate(hs)
        want = O.unique_sort_hashes(allh) if len(allh) else np.zeros((0, 2), np.int32)
        assert np.array_equal(res.clip_hashes(c), want), (seed, c)
    for u, pk in enumerate(unit_peaks):
        assert np.array_equal(lms[u], O.peaks2landmarks(pk, prm).astype(np.int32)), (seed, u)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(10))
def test_gpu_lane_per_peak_kernel_class(seed):
    """The parameter class k_pairlane serves (one shift, |df| window <= 63 bins, dt field not wrapping,
    <= 8 peaks per column, fanout <= 8): dense columns (several 64-peak rounds per wavefront, columns
    straddling rounds), empty units, units shorter than mindt; checked against the oracle and against
    k_pairmerge on the same input."""
    from audfprint_amd.batch import Extractor
    rng = np.random.RandomState(7000 + seed)
    kw = dict(maxpairsperpeak=int(rng.randint(1, 9)), targetdf=int(rng.randint(1, 33)), mindt=int(rng.randint(0, 5)),
              targetdt=int(rng.randint(1, 65)), maxpksperframe=int(rng.randint(1, 9)), shifts=1)
    prm = O.Params(**kw)
    lens = [0, 1, int(rng.randint(2, 6)), 63, 64, 65, 256, 257, int(rng.randint(300, 900)), int(rng.randint(1000, 1500))]
    dens = [float(rng.choice([0.05, 0.5, 2.0, 7.9])) for _ in lens]
    unit_peaks = [_random_peaks(rng, n, d, kw['maxpksperframe']) for n, d in zip(lens, dens)]
    want = []
    for pk in unit_peaks:
        h = O.landmarks2hashes(O.peaks2landmarks(pk, prm))
        want.append(O.unique_sort_hashes(h) if len(h) else np.zeros((0, 2), np.int32))
    got = {}
    for name, env in (('lane', None), ('merge', 'AFP_NO_PAIRLANE')):
        if env:
            os.environ[env] = '1'
        try:
            e = Extractor(0)
        finally:
            if env:
                os.environ.pop(env, None)
        e.set_params(**kw)
        res, _ = e.pairs_from_peaks(unit_peaks, want_hashes=True, want_landmarks=False)
        got[name] = res
        e.close()
        for c in range(len(unit_peaks)):
            assert np.array_equal(res.clip_hashes(c), want[c]), (name, seed, c, kw)
    assert np.array_equal(got['lane'].hashes, got['merge'].hashes)


@pytest.mark.gpu
@pytest.mark.parametrize('flavour', ['lane', 'merge', 'generic'])
def test_gpu_more_peaks_per_column_than_maxpksperframe(flavour):
    """A .afpk written with another maxpksperframe (or by hand) may hold more peaks in one column than the
    analyzer's own setting; peaks2landmarks (audfprint_analyze.py:310-343) pairs them all."""
    from audfprint_amd.batch import Extractor
    env = {'lane': None, 'merge': 'AFP_NO_PAIRLANE', 'generic': 'AFP_GENERIC_PAIR'}[flavour]
    if env:
        os.environ[env] = '1'
    try:
        e = Extractor(0)
    finally:
        if env:
            os.environ.pop(env, None)
    rng = np.random.RandomState(4242)
    for maxk_in, kset in ((7, 5), (12, 5), (40, 3)):
        kw = dict(maxpksperframe=kset, maxpairsperpeak=3, shifts=1)
        prm = O.Params(**kw)
        e.set_params(**kw)
        unit_peaks = [_random_peaks(rng, 150, maxk_in * 0.8, maxk_in) for _ in range(3)]
        assert max(np.bincount(pk[:, 0]).max() for pk in unit_peaks) > kset
        res, lms = e.pairs_from_peaks(unit_peaks, want_hashes=True, want_landmarks=True)
        for c, pk in enumerate(unit_peaks):
            lm = O.peaks2landmarks(pk, prm)
            assert np.array_equal(lms[c], lm.astype(np.int32)), (flavour, maxk_in, c)
            assert np.array_equal(res.clip_hashes(c), O.unique_sort_hashes(O.landmarks2hashes(lm))), (flavour, maxk_in, c)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(14))
def test_gpu_lane_per_peak_multi_shift_kernel_class(seed):
    """The parameter class k_pairlane_ms serves (2..8 shifts, |df| window <= 63 bins, dt field not wrapping, <= 8 peaks per
    (shift, column), fanout <= 16): its round loop was rewritten in round 4 (hits counted in a lockstep pass over the target
    frames, remembered as 6-bit entries -- a second register beyond ten entries, i.e. fanout > 10 -- then emitted hit by hit
    into one list per round).  Dense columns (more than 64 and more than 128 hashes per merged column), sparse ones, empty
    units, units of different lengths inside a clip, several rounds per wavefront; against the oracle and against
    k_pairmerge on the same input."""
    from audfprint_amd.batch import Extractor
    rng = np.random.RandomState(9100 + seed)
    S = int(rng.choice([2, 3, 4, 4, 8]))
    K = int(rng.randint(1, 9))
    kw = dict(maxpairsperpeak=int(rng.choice([1, 3, 10, 11, 16])) if seed % 3 else int(rng.randint(1, 17)),
              targetdf=int(rng.randint(1, 33)), mindt=int(rng.randint(0, 5)), targetdt=int(rng.randint(2, 65)),
              maxpksperframe=K, shifts=S)
    prm = O.Params(**kw)
    nclips = 6
    base = [0, int(rng.randint(1, 6)), 64, int(rng.randint(65, 300)), int(rng.randint(300, 700)), int(rng.randint(700, 1300))]
    unit_peaks = []
    for c in range(nclips):
        d = float(rng.choice([0.05, 0.6, 2.5, 7.9]))
        for s in range(S):
            n = max(0, base[c] - int(rng.randint(0, 2)))          # the shifts of a clip differ by at most a frame
            unit_peaks.append(_random_peaks(rng, n, d, K))
    want = []
    for c in range(nclips):
        hs = [O.landmarks2hashes(O.peaks2landmarks(unit_peaks[c * S + s], prm)) for s in range(S)]
        allh = np.concatenate(hs)
        want.append(O.unique_sort_hashes(allh) if len(allh) else np.zeros((0, 2), np.int32))
    got = {}
    for name, env in (('lane_ms', None), ('merge', 'AFP_NO_PAIRLANE')):
        if env:
            os.environ[env] = '1'
        try:
            e = Extractor(0)
        finally:
            if env:
                os.environ.pop(env, None)
        e.set_params(**kw)
        res, _ = e.pairs_from_peaks(unit_peaks, want_hashes=True, want_landmarks=False)
        got[name] = res
        e.close()
        for c in range(nclips):
            assert np.array_equal(res.clip_hashes(c), want[c]), (name, seed, c, kw)
    assert np.array_equal(got['lane_ms'].hashes, got['merge'].hashes)


def _list_order_peaks(rng, nframes, density, maxk, dup_p):
    """Peak lists in orders find_peaks never emits but Analyzer.peaks2landmarks accepts: bins shuffled inside a column, bins
    listed twice, columns out of order (the last row holds the largest column)."""
    rows = []
    for t in range(nframes):
        k = min(maxk, rng.poisson(density))
        bins = [int(b) for b in rng.choice(256, size=k, replace=True)]
        bins += [b for b in bins if rng.rand() < dup_p]
        rng.shuffle(bins)
        rows += [(t, b) for b in bins]
    rows = np.array(rows, dtype=np.int32).reshape(-1, 2)
    if len(rows) > 2 and rng.rand() < 0.5:
        # move whole rows around; keep the last row (largest column) last -- per-column list order is what is left of it
        perm = rng.permutation(len(rows) - 1)
        rows = np.concatenate([rows[perm], rows[-1:]])
    return rows


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')
@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), nframes=st.integers(1, 120), density=st.floats(0.05, 3.0),
       fanout=st.integers(1, 12), targetdf=st.integers(1, 40), mindt=st.integers(0, 4), targetdt=st.integers(1, 70))
def test_oracle_pairing_equals_reference_on_list_order_peaks(seed, nframes, density, fanout, targetdf, mindt, targetdt):
    sys.path.insert(0, REF)
    try:
        import audfprint_analyze as R
    finally:
        sys.path.remove(REF)
    rng = np.random.RandomState(seed)
    pk = _list_order_peaks(rng, nframes, density, 6, 0.2)
    an = R.Analyzer()
    an.maxpairsperpeak, an.targetdf, an.mindt, an.targetdt = fanout, targetdf, mindt, targetdt
    prm = O.Params(maxpairsperpeak=fanout, targetdf=targetdf, mindt=mindt, targetdt=targetdt)
    ref_lm = an.peaks2landmarks([(int(c), int(b)) for c, b in pk])
    got_lm = O.peaks2landmarks(pk, prm)
    assert np.array_equal(np.array(ref_lm, dtype=np.int64).reshape(-1, 4), got_lm)
    assert np.array_equal(R.landmarks2hashes(ref_lm), O.landmarks2hashes(got_lm))


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(8))
def test_gpu_pairing_of_list_order_peak_lists(seed):
    """Analyzer.peaks2landmarks (audfprint_analyze.py:321-341) follows the order of the rows inside a column and pairs a bin
    as often as it is listed; k_pair_rows does the same (landmarks in the reference's order, hashes sorted unique per clip
    over all shifts).  Mixed batches: one list-order unit sends the whole call down the row path."""
    from audfprint_amd.batch import Extractor
    rng = np.random.RandomState(5100 + seed)
    kw = dict(maxpairsperpeak=int(rng.choice([1, 3, 10, 25])), targetdf=int(rng.choice([5, 31, 40])),
              mindt=int(rng.choice([0, 1, 2])), targetdt=int(rng.choice([8, 63, 100])),
              maxpksperframe=int(rng.choice([3, 5, 9])), shifts=int(rng.choice([1, 1, 2, 4])))
    prm = O.Params(**kw)
    e = Extractor(0)
    e.set_params(**kw)
    nclips = 5
    unit_peaks = []
    for u in range(nclips * prm.shifts):
        n = int(rng.choice([0, 1, 3, 70, 300, 520]))
        if u % 3 == 2:
            unit_peaks.append(_random_peaks(rng, n, float(rng.uniform(0.1, 2.5)), kw['maxpksperframe']))
        else:
            unit_peaks.append(_list_order_peaks(rng, n, float(rng.uniform(0.1, 3.5)), 12, float(rng.choice([0.0, 0.3]))))
    res, lms = e.pairs_from_peaks(unit_peaks, want_hashes=True, want_landmarks=True)
    e.close()
    for u, pk in enumerate(unit_peaks):
        assert np.array_equal(lms[u], O.peaks2landmarks(pk, prm).astype(np.int32)), (seed, u)
    for c in range(nclips):
        hs = [O.landmarks2hashes(O.peaks2landmarks(unit_peaks[c * prm.shifts + s], prm)) for s in range(prm.shifts)]
        allh = np.concatenate(hs)
        want = O.unique_sort_hashes(allh) if len(allh) else np.zeros((0, 2), np.int32)
        assert np.array_equal(res.clip_hashes(c), want), (seed, c)
