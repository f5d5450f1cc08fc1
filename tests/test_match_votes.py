"""Vote counting of the matcher (second half): Matcher._best_count_ids /
_approx_match_counts / match_hashes (audfprint_match.py:124-147, 241-352).
CPU: the oracle restatement against the golden fixture made from the live reference
(tests/golden/make_golden_match.py) and against the live reference on random hit lists; the host half of
audfprint_amd.match on a stand-in for the device.  GPU: audfprint_amd.match through the C ABI."""
import os
import random
import sys

import numpy as np
import pytest

from oracle import afp_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


MODES = ((1, 0), (1, 1), (0, 1))          # (exact_count, find_time_range) beyond the default switches


def _gold():
    z = np.load(os.path.join(HERE, 'golden', 'match_votes.npz'))
    return z, [str(n) for n in z['names']]


def _oracle_table(z, names):
    ht = O.OracleHashTable(hashbits=20, depth=100)
    rng = random.Random(4321)
    off = z['offsets']
    for i, nm in enumerate(names):
        ht.store(nm, z['rows'][off[i]:off[i + 1]], rng)
    return ht


def _settings(z):
    for si in range(int(z['nsettings'])):
        w, th, sd, ma = [int(v) for v in z['set%d' % si]]
        yield si, dict(window=w, threshcount=th, search_depth=sd, max_alignments_per_id=ma)


def test_oracle_vote_counting_equals_reference_golden():
    z, names = _gold()
    ht = _oracle_table(z, names)
    for si, kw in _settings(z):
        for qi in range(int(z['nqueries'])):
            q = z['q%d' % qi]
            hits = ht.get_hits(q)
            ids, raw = O.match_best_count_ids(hits, ht.hashesperid, kw['threshcount'], kw['search_depth'])
            assert np.array_equal(ids, z['s%d_q%d_ids' % (si, qi)]) and np.array_equal(raw, z['s%d_q%d_raw' % (si, qi)])
            assert np.array_equal(O.match_hashes(ht, q, **kw), z['s%d_q%d_res' % (si, qi)]), (si, qi)
            for ec, tr in MODES:        # exact counts (:195-239), time ranges (:173-193)
                got = O.match_hashes(ht, q, exact_count=bool(ec), find_time_range=bool(tr), **kw)
                assert np.array_equal(got, z['s%d_q%d_res_e%d_t%d' % (si, qi, ec, tr)]), (si, qi, ec, tr)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')
@pytest.mark.parametrize('seed', range(8))
def test_oracle_vote_counting_equals_live_reference_on_random_hits(seed):
    sys.path.insert(0, REF)
    try:
        import audfprint_match as RM
    finally:
        sys.path.remove(REF)
    rng = np.random.RandomState(300 + seed)
    nids = int(rng.randint(1, 40))
    n = int(rng.randint(1, 4000))
    ids = rng.randint(0, nids, n)
    # a few ids get a concentrated skew (a "match"), the rest are spread out
    skew = np.where(rng.rand(n) < 0.4, (ids * 37) % 200 - 100 + rng.randint(-1, 2, n), rng.randint(-3000, 3000, n))
    hits = np.stack([ids, skew, rng.randint(0, 1 << 20, n), rng.randint(0, 500, n)], axis=1).astype(np.int32)

    class HT(object):
        hashesperid = rng.randint(1, 3000, nids).astype(np.uint32)

    m = RM.Matcher()
    m.window = int(rng.randint(0, 4)); m.threshcount = int(rng.randint(0, 8))
    m.search_depth = int(rng.randint(1, 50)); m.max_alignments_per_id = int(rng.randint(0, 5))
    rid, rraw = m._best_count_ids(hits, HT)
    oid, oraw = O.match_best_count_ids(hits, HT.hashesperid, m.threshcount, m.search_depth)
    assert np.array_equal(rid, oid) and np.array_equal(rraw, oraw)
    assert np.array_equal(m._approx_match_counts(hits, rid, rraw),
                          O.match_approx_counts(hits, oid, oraw, m.window, m.threshcount, m.max_alignments_per_id))
    if m.threshcount >= 1:
        # the remaining modes on the same hit list (a threshold of 0 lets the reference index an empty time list: it raises)
        for tr in (False, True):
            m.find_time_range = tr
            assert np.array_equal(m._exact_match_counts(hits, rid, rraw),
                                  O.match_exact_counts(hits, oid, oraw, m.window, m.threshcount, tr, m.time_quantile)), tr


class _FakeDevice(object):
    """Stands in for the C ABI calls of VoteCounter: same outputs, computed with numpy from the hit rows."""

    def __init__(self, hits):
        self.hits_, self.nhits = hits, len(hits)

    def id_counts(self):
        ids = np.unique(self.hits_[:, 0])
        return ids.astype(np.int32), np.bincount(self.hits_[:, 0])[ids].astype(np.int32) if len(ids) else np.zeros(0, np.int32)

    def max_orig_time(self):
        return int(self.hits_[:, 3].max()) if len(self.hits_) else 0

    def select(self, ids, lo, hi):
        h = self.hits_
        return [h[(h[:, 0] == i) & (h[:, 1] >= a) & (h[:, 1] <= b)][:, [3, 2]].astype(np.int32) for i, a, b in zip(ids, lo, hi)]

    def skew_hist(self, ids):
        mt = int(self.hits_[:, 1].min())
        width = int(self.hits_[:, 1].max()) - mt + 1
        hist = np.zeros((len(ids), width), np.int32)
        for r, i in enumerate(ids):
            sel = self.hits_[self.hits_[:, 0] == i, 1] - mt
            hist[r] = np.bincount(sel, minlength=width)
        return mt, hist


def test_host_half_of_vote_counter_on_golden_hits():
    """The numpy half of audfprint_amd.match (weighting / argsort / mode picking) with the device replaced."""
    from audfprint_amd import match as M
    z, names = _gold()
    ht = _oracle_table(z, names)
    for si, kw in _settings(z):
        for qi in range(int(z['nqueries'])):
            hits = ht.get_hits(z['q%d' % qi])
            vc = M.VoteCounter.__new__(M.VoteCounter)
            fake = _FakeDevice(hits)
            vc.nhits, vc.id_counts, vc.skew_hist = fake.nhits, fake.id_counts, fake.skew_hist
            vc.select, vc.max_orig_time = fake.select, fake.max_orig_time
            ids, raw = vc.best_count_ids(ht.hashesperid, kw['threshcount'], kw['search_depth'])
            assert np.array_equal(ids, z['s%d_q%d_ids' % (si, qi)]) and np.array_equal(raw, z['s%d_q%d_raw' % (si, qi)])
            res = vc.approx_match_counts(ids, raw, kw['window'], kw['threshcount'], kw['max_alignments_per_id'])
            res = res[(-res[:, 1]).argsort(), ]
            assert np.array_equal(res, z['s%d_q%d_res' % (si, qi)]), (si, qi)
            for ec, tr in MODES:
                if ec:
                    res = vc.exact_match_counts(ids, raw, kw['window'], kw['threshcount'], bool(tr), 0.02)
                else:
                    res = vc.approx_match_counts(ids, raw, kw['window'], kw['threshcount'], kw['max_alignments_per_id'], True, 0.02)
                res = res[(-res[:, 1]).argsort(), ]
                assert np.array_equal(res, z['s%d_q%d_res_e%d_t%d' % (si, qi, ec, tr)]), (si, qi, ec, tr)


def test_hashesfor_rows_from_a_device_selection_equal_the_reference():
    """Matcher.match_hashes(..., hashesfor=k) (audfprint_match.py:346-352) -> _unique_match_hashes (:149-171): the host half of
    VoteCounter.unique_match_hashes on a stand-in device, against the oracle restatement and -- where the tree is mounted -- the
    live reference's own method, on the golden queries, for every result row."""
    from audfprint_amd import match as M
    z, names = _gold()
    ht = _oracle_table(z, names)
    RM = None
    if os.path.isdir(REF):
        sys.path.insert(0, REF)
        try:
            import audfprint_match as RM
        finally:
            sys.path.remove(REF)
    checked = 0
    for qi in range(int(z['nqueries'])):
        hits = ht.get_hits(z['q%d' % qi])
        if not len(hits):
            continue
        vc = M.VoteCounter.__new__(M.VoteCounter)
        fake = _FakeDevice(hits)
        vc.nhits, vc.select, vc.max_orig_time = fake.nhits, fake.select, fake.max_orig_time
        res = z['s0_q%d_res' % qi]
        for k in range(min(3, len(res))):
            for window in (0, 1, 3):
                got = vc.unique_match_hashes(res[k, 0], res[k, 2], window)
                want = O.match_unique_hashes(hits, res[k, 0], res[k, 2], window)
                assert got.dtype == want.dtype and np.array_equal(got, want), (qi, k, window)
                if RM is not None:
                    m = RM.Matcher()
                    m.window = window
                    assert np.array_equal(m._unique_match_hashes(res[k, 0], hits, res[k, 2]), want), (qi, k, window)
                checked += 1
    assert checked >= 9


class _Matcher(object):
    """The attribute surface of audfprint_match.Matcher that match_hashes reads (:95-122)."""
    window, threshcount, search_depth, max_alignments_per_id = 1, 5, 100, 100
    exact_count = False
    find_time_range = False
    time_quantile = 0.02


@pytest.mark.gpu
def test_gpu_match_hashes_equals_reference_golden():
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    from audfprint_amd import match as M
    z, names = _gold()
    ht = O.OracleHashTable(hashbits=20, depth=100)
    tb = TableBuilder(ht, Extractor.get(0))
    random.seed(4321)
    tb.store_batch(names, rows=z['rows'], offsets=z['offsets'])
    for si, kw in _settings(z):
        m = _Matcher()
        for k, v in kw.items():
            setattr(m, k, v)
        for qi in range(int(z['nqueries'])):
            q = z['q%d' % qi]
            vc = M.VoteCounter(tb)
            vc.query(q)
            ids, raw = vc.best_count_ids(ht.hashesperid, m.threshcount, m.search_depth)
            assert np.array_equal(ids, z['s%d_q%d_ids' % (si, qi)]) and np.array_equal(raw, z['s%d_q%d_raw' % (si, qi)])
            res = M.match_hashes(m, tb, q)
            assert res.dtype == np.int32 and np.array_equal(res, z['s%d_q%d_res' % (si, qi)]), (si, qi)
            # exact counts and time ranges from the rows selected on the device (afp_table_select_hits)
            for ec, tr in MODES:
                m.exact_count, m.find_time_range = bool(ec), bool(tr)
                res = M.match_hashes(m, tb, q)
                assert res.dtype == np.int32 and np.array_equal(res, z['s%d_q%d_res_e%d_t%d' % (si, qi, ec, tr)]), (si, qi, ec, tr)
            m.exact_count = m.find_time_range = False
            # hashesfor (:346-352): the matching hashes of result k from a selection on the device
            res, mh = M.match_hashes(m, tb, q, hashesfor=0) if len(z['s%d_q%d_res' % (si, qi)]) else (None, None)
            if res is not None:
                tb.finalize()
                want = O.match_unique_hashes(ht.get_hits(q), res[0, 0], res[0, 2], m.window)
                assert np.array_equal(mh, want) and len(mh) > 0, (si, qi)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(4))
def test_gpu_vote_histograms_on_random_tables(seed):
    """id counts and per-id skew histograms against numpy on the downloaded hit rows; small table with
    overflowing buckets, many ids, negative and large skews."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    from audfprint_amd import match as M
    rng = np.random.RandomState(900 + seed)
    hashbits, depth = int(rng.choice([8, 12])), int(rng.choice([3, 20]))
    ht = O.OracleHashTable(hashbits=hashbits, depth=depth)
    tb = TableBuilder(ht, Extractor.get(0))
    ntr = int(rng.randint(1, 300))
    rows, off, names = [], [0], []
    for t in range(ntr):
        n = int(rng.randint(0, 60))
        r = np.stack([np.sort(rng.randint(0, 16000, n)), rng.randint(0, 1 << hashbits, n)], axis=1).astype(np.int32)
        rows.append(r); off.append(off[-1] + n); names.append('t%d' % t)
    random.seed(5)
    tb.store_batch(names, rows=np.concatenate(rows).reshape(-1, 2), offsets=np.array(off, np.int64))
    q = np.stack([rng.randint(0, 16000, 500), rng.randint(0, 1 << hashbits, 500)], axis=1).astype(np.int32)
    vc = M.VoteCounter(tb)
    nh = vc.query(q)
    hits = vc.hits()
    assert len(hits) == nh
    ids, cnt = vc.id_counts()
    if nh == 0:
        assert len(ids) == 0
        return
    assert np.array_equal(ids, np.unique(hits[:, 0])) and np.array_equal(cnt, np.bincount(hits[:, 0])[ids])
    want = ids[rng.permutation(len(ids))[:min(len(ids), 37)]]
    mt, hist = vc.skew_hist(want)
    assert mt == hits[:, 1].min() and hist.shape == (len(want), hits[:, 1].max() - mt + 1)
    for r, i in enumerate(want):
        assert np.array_equal(hist[r], np.bincount(hits[hits[:, 0] == i, 1] - mt, minlength=hist.shape[1])), (seed, i)
    # an empty query after a non-empty one
    assert vc.query(np.zeros((0, 2), np.int32)) == 0
    ids0, cnt0 = vc.id_counts()
    assert len(ids0) == 0 and len(cnt0) == 0
    # the row selection behind exact counts / time ranges: arbitrary (id, skew range) queries -- repeated ids, empty ranges, ids
    # without hits -- against numpy on the downloaded rows (row order inside a query is unspecified: compared sorted)
    vc.query(q)
    vc.id_counts()
    assert vc.max_orig_time() == hits[:, 3].max()
    qi = np.concatenate([ids[rng.randint(0, len(ids), 40)], [int(ids.max()) + 1, 0]]).astype(np.int32)
    lo = rng.randint(int(hits[:, 1].min()) - 5, int(hits[:, 1].max()) + 5, len(qi)).astype(np.int32)
    hi = (lo + rng.randint(-2, 400, len(qi))).astype(np.int32)
    sel = vc.select(qi, lo, hi)
    for k in range(len(qi)):
        want_rows = hits[(hits[:, 0] == qi[k]) & (hits[:, 1] >= lo[k]) & (hits[:, 1] <= hi[k])][:, [3, 2]]
        got_rows = sel[k]
        assert got_rows.shape == want_rows.shape, (seed, k)
        assert np.array_equal(got_rows[np.lexsort((got_rows[:, 1], got_rows[:, 0]))], want_rows[np.lexsort((want_rows[:, 1], want_rows[:, 0]))])
    assert vc.select([], [], []) == []
    m = _Matcher()
    assert M.match_hashes(m, tb, np.zeros((0, 2), np.int32)).shape == (0, 7)
    m.exact_count = m.find_time_range = True
    assert M.match_hashes(m, tb, np.zeros((0, 2), np.int32)).shape == (0, 7)
