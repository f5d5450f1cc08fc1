"""batch hash-table build.  CPU: the oracle's restatement of HashTable.store
vs the golden table made by the live reference; GPU: TableBuilder vs the same golden (bit-exact,
including the bucket-overflow path replayed with Python's seeded `random`)."""
import os
import random

import numpy as np
import pytest

from oracle import afp_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'table_store.npz')


def _gold():
    z = np.load(GOLD)
    names = [str(n) for n in z['names']]
    return z, names


def test_oracle_store_equals_reference_table():
    z, names = _gold()
    off = z['offsets']
    for hashbits, depth, key in ((10, 4, 'small'), (20, 100, 'big')):
        ht = O.OracleHashTable(hashbits=hashbits, depth=depth)
        rng = random.Random(1234)
        for i, nm in enumerate(names):
            ht.store(nm, z['rows'][off[i]:off[i + 1]], rng)
        if key == 'small':
            assert np.array_equal(ht.table, z['small_table']) and np.array_equal(ht.counts, z['small_counts'])
            assert np.array_equal(ht.hashesperid, z['small_hpi']) and ht.names == [str(n) for n in z['small_names']]
        else:
            b = z['big_buckets']
            assert np.array_equal(np.nonzero(ht.counts)[0], b)
            assert np.array_equal(ht.table[b], z['big_rows']) and np.array_equal(ht.counts[b], z['big_counts'])
            assert np.array_equal(ht.hashesperid, z['big_hpi'])


def test_oracle_store_fast_is_store_row_for_row():
    """bench.py checks the 12 500-clip job's table with OracleHashTable.store_fast (the loop of store() batched per call):
    it must BE store() -- same table, counts, hashesperid and the same generator state afterwards -- on the reference golden
    (small table: most rows meet full buckets), on a clip stored twice (the name is re-used, hash_table.py:325-344) and on
    random rows with popular hashes and a bucket that fills up INSIDE one call."""
    z, names = _gold()
    off = z['offsets']
    for hashbits, depth in ((10, 4), (12, 7), (20, 100)):
        a, b = O.OracleHashTable(hashbits=hashbits, depth=depth), O.OracleHashTable(hashbits=hashbits, depth=depth)
        ra, rb = random.Random(1234), random.Random(1234)
        seq = list(range(len(names))) + [0, 3]
        for i in seq:
            a.store(names[i], z['rows'][off[i]:off[i + 1]], ra)
            b.store_fast(names[i], z['rows'][off[i]:off[i + 1]], rb)
        rng = np.random.RandomState(9)
        for j in range(6):
            rows = np.stack([rng.randint(0, 40000, 3000), rng.randint(0, 1 << 22, 3000)], 1).astype(np.int32)
            rows[:900, 1] = rng.randint(0, 5, 900)                   # 180 rows per popular hash in one call
            a.store('x%d' % j, rows, ra)
            b.store_fast('x%d' % j, rows, rb)
        a.store('empty', np.zeros((0, 2), np.int32), ra)
        b.store_fast('empty', np.zeros((0, 2), np.int32), rb)
        assert np.array_equal(a.table, b.table) and np.array_equal(a.counts, b.counts)
        assert np.array_equal(a.hashesperid, b.hashesperid) and a.names == b.names
        assert ra.getstate() == rb.getstate() and int(np.sum(a.counts > depth)) > 0
    zt = z['small_table']
    c = O.OracleHashTable(hashbits=10, depth=4)
    rc = random.Random(1234)
    for i, nm in enumerate(names):
        c.store_fast(nm, z['rows'][off[i]:off[i + 1]], rc)
    assert np.array_equal(c.table, zt) and np.array_equal(c.counts, z['small_counts'])      # ... and the live reference's golden


@pytest.mark.gpu
@pytest.mark.parametrize('split', [None, 3, 1])
def test_gpu_table_build_small_with_overflow(split):
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    z, names = _gold()
    off = z['offsets']
    ht = O.OracleHashTable(hashbits=10, depth=4)              # duck-typed container with name_to_id
    tb = TableBuilder(ht, Extractor.get(0))
    random.seed(1234)
    cuts = [0, len(names)] if split is None else list(range(0, len(names), split)) + [len(names)]
    total_ovf = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        total_ovf += tb.store_batch(names[a:b], rows=z['rows'][off[a]:off[b]], offsets=off[a:b + 1] - off[a])
    tb.finalize()
    assert total_ovf == int(np.sum(np.maximum(z['small_counts'] - 4, 0)))
    assert np.array_equal(ht.counts, z['small_counts'])
    assert np.array_equal(ht.table, z['small_table'])
    assert np.array_equal(ht.hashesperid, z['small_hpi']) and ht.names == [str(n) for n in z['small_names']]


@pytest.mark.gpu
def test_gpu_table_build_from_device_resident_hashes():
    """extract -> store without the hashes ever leaving HBM; default-size table, no overflow."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    z, names = _gold()
    ex = Extractor.get(0)
    ex.set_params()
    clips = [O.synth_noise(9000 + i, 4.0) for i in range(6)]
    ht = O.OracleHashTable(hashbits=20, depth=100)
    tb = TableBuilder(ht, ex)
    r = ex.extract(clips=clips, want_hashes=True)
    assert np.array_equal(r.hashes, z['rows'])
    assert tb.store_batch(names, offsets=r.hash_offsets) == 0
    tb.finalize()
    b = z['big_buckets']
    assert np.array_equal(np.nonzero(ht.counts)[0], b)
    assert np.array_equal(ht.table[b], z['big_rows']) and np.array_equal(ht.counts[b], z['big_counts'])
    assert np.array_equal(ht.hashesperid, z['big_hpi'])


@pytest.mark.gpu
def test_gpu_table_popular_bucket_and_existing_table():
    """A bucket that receives hundreds of rows in one batch (long-segment kernel) on top of a table
    that already holds entries; compared with the oracle's sequential store."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    rng = np.random.RandomState(5)
    nclips = 40
    rows, off = [], [0]
    for c in range(nclips):
        n = int(rng.randint(50, 200))
        t = np.sort(rng.randint(0, 300, n))
        h = np.where(rng.rand(n) < 0.3, 777, rng.randint(0, 1 << 12, n))     # hash 777 is very popular
        rows.append(np.stack([t, h], axis=1))
        off.append(off[-1] + n)
    rows = np.concatenate(rows).astype(np.int32)
    off = np.array(off, np.int64)
    names = ['f%d' % i for i in range(nclips)]
    ref = O.OracleHashTable(hashbits=12, depth=600)
    prng = random.Random(99)
    for i in range(nclips):
        ref.store(names[i], rows[off[i]:off[i + 1]], prng)
    ht = O.OracleHashTable(hashbits=12, depth=600)
    prng2 = random.Random(99)
    for i in range(10):                                       # first 10 clips by the sequential loop ...
        ht.store(names[i], rows[off[i]:off[i + 1]], prng2)
    tb = TableBuilder(ht, Extractor.get(0))                   # ... the rest on the GPU, on top of that table
    random.setstate(prng2.getstate())
    tb.store_batch(names[10:], rows=rows[off[10]:], offsets=off[10:] - off[10])
    tb.finalize()
    assert np.array_equal(ht.counts, ref.counts) and np.array_equal(ht.table, ref.table)
    assert np.array_equal(ht.hashesperid, ref.hashesperid)


def test_oracle_get_hits_equals_reference():
    z, names = _gold()
    off = z['offsets']
    for hashbits, depth, key in ((10, 4, 'small_hits'), (20, 100, 'big_hits')):
        ht = O.OracleHashTable(hashbits=hashbits, depth=depth)
        rng = random.Random(1234)
        for i, nm in enumerate(names):
            ht.store(nm, z['rows'][off[i]:off[i + 1]], rng)
        assert np.array_equal(ht.get_hits(z['q_rows']), z[key])


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [(10, 4, 'small_hits'), (20, 100, 'big_hits')])
def test_gpu_get_hits(cfg):
    """Build on the GPU (overflow replay included), then query the device-resident table."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    hashbits, depth, key = cfg
    z, names = _gold()
    off = z['offsets']
    ht = O.OracleHashTable(hashbits=hashbits, depth=depth)
    tb = TableBuilder(ht, Extractor.get(0))
    random.seed(1234)
    tb.store_batch(names, rows=z['rows'], offsets=off)
    hits = tb.get_hits(z['q_rows'])
    assert hits.dtype == np.int32 and np.array_equal(hits, z[key])
    assert tb.get_hits(np.zeros((0, 2), np.int32)).shape == (0, 4)


def test_native_randint_replay_is_this_interpreters_stream():
    """afp_mt_randint_replay (host function of the library, no GPU needed): random.randint(0, n) exactly as CPython draws
    it -- values AND generator state, across state regenerations and every bit length -- because the reference's
    HashTable.store draws from the global generator (hash_table.py:128) and later draws must continue unchanged."""
    import ctypes as C
    from audfprint_amd import _lib, table
    lib = _lib.load()
    assert table.native_randint_ok(lib)
    g = random.Random(20240917)
    for _ in range(700):
        g.random()                                   # (an arbitrary position inside the state block)
    st = g.getstate()
    rs = np.random.RandomState(3)
    counts = np.concatenate([rs.randint(0, 1 << int(b), size=300) for b in range(1, 32)] +
                            [np.array([0, 1, 2, 3, 4, 99, 100, 101, (1 << 31) - 1, (1 << 30), (1 << 30) - 1])]).astype(np.int32)
    want = [g.randint(0, int(c)) for c in counts]
    mt = np.array(st[1][:624], dtype=np.uint32)
    pos = C.c_int32(st[1][624])
    out = np.empty(len(counts), np.int32)
    assert lib.afp_mt_randint_replay(mt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                     len(counts), out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert out.tolist() == want
    # hand the state back the way TableBuilder does and keep drawing: the two generators stay in step
    g2 = random.Random()
    g2.setstate((st[0], tuple(mt.tolist()) + (int(pos.value),), st[2]))
    assert [g2.random() for _ in range(5)] == [g.random() for _ in range(5)] and g2.getstate() == g.getstate()
    bad = np.array([-1], np.int32)
    assert lib.afp_mt_randint_replay(mt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), bad.ctypes.data_as(C.POINTER(C.c_int32)), 1,
                                     out.ctypes.data_as(C.POINTER(C.c_int32))) < 0


@pytest.mark.gpu
def test_gpu_overflow_replay_native_equals_the_python_loop_and_leaves_random_in_step(monkeypatch):
    """The overflow draws made by the library (afp_table_replay_overflow) and by Python's own random.randint give the same
    table AND leave the global generator in the same state (the next random.random() is the same number)."""
    from audfprint_amd import table as T
    from audfprint_amd.batch import Extractor
    z, names = _gold()
    off = z['offsets']
    got = []
    for native in (True, False):
        monkeypatch.setattr(T, '_native_randint', native)
        ht = O.OracleHashTable(hashbits=10, depth=4)
        tb = T.TableBuilder(ht, Extractor.get(0))
        random.seed(1234)
        novf = 0
        for a, b in ((0, 2), (2, 3), (3, len(names))):
            novf += tb.store_batch(names[a:b], rows=z['rows'][off[a]:off[b]], offsets=off[a:b + 1] - off[a])
        tb.finalize()
        got.append((novf, ht.table.copy(), ht.counts.copy(), random.random()))
    monkeypatch.setattr(T, '_native_randint', None)
    assert got[0][0] == got[1][0] > 0
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2]) and got[0][3] == got[1][3]
    assert np.array_equal(got[0][1], z['small_table']) and np.array_equal(got[0][2], z['small_counts'])


@pytest.mark.gpu
def test_gpu_several_contexts_feed_one_table_in_clip_order():
    """store_batch(src=ctx): rows of OTHER contexts' batches, still in HBM, go into one table in the caller's order --
    the shape of the pipelined `new` job (bench.py c4_job).  Equal to the oracle's sequential store with the same seed."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    ex = Extractor.get(0)
    ctxs = [Extractor(0), Extractor(0)]
    try:
        clips = [O.synth_noise(9100 + i, 3.0 + 0.25 * (i % 5)) for i in range(18)]
        names = ['c%02d' % i for i in range(len(clips))]
        want = O.OracleHashTable(hashbits=9, depth=6)           # small: plenty of full buckets
        rr = random.Random(77)
        for nm, d in zip(names, clips):
            want.store(nm, O.extract(d, O.Params())[1], rr)
        ht = O.OracleHashTable(hashbits=9, depth=6)
        tb = TableBuilder(ht, ex)
        random.seed(77)
        batches = [(0, 6), (6, 12), (12, 18)]
        pend = []
        for k, (a, b) in enumerate(batches):
            c = ctxs[k % 2]
            c.set_params()
            if len(pend) == 2:
                c0, a0, b0 = pend.pop(0)
                tb.store_batch(names[a0:b0], offsets=c0.fetch(b0 - a0, True, False).hash_offsets, src=c0)
            pcm, off = Extractor.pack(clips[a:b])
            c._keep = (pcm, off)
            c.submit(pcm, off)
            pend.append((c, a, b))
        for c0, a0, b0 in pend:
            tb.store_batch(names[a0:b0], offsets=c0.fetch(b0 - a0, True, False).hash_offsets, src=c0)
        tb.finalize()
        assert int(np.sum(want.counts > 6)) > 0
        assert np.array_equal(ht.counts, want.counts) and np.array_equal(ht.table, want.table)
        assert np.array_equal(ht.hashesperid, want.hashesperid) and ht.names == want.names
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_gpu_large_uploads_of_several_contexts_share_the_upload_stream():
    """Uploads of 4 MB and more leave the handle's stream for the device's upload stream (afp_abi.hip upload_stream()); the
    consumer kernels are ordered behind them by an event.  Three contexts, each re-used while the others are in flight,
    s16 and float32, pageable memory: every batch equals what extract() returns for the same clips."""
    from audfprint_amd.batch import Extractor
    ctxs = [Extractor(0) for _ in range(3)]
    ref = Extractor.get(0)
    ref.set_params()
    try:
        base = [O.synth_noise(9300 + i, 10.0) for i in range(8)]
        batches = []
        for k in range(7):
            clips = [base[(k + j) % len(base)] for j in range(24 + k)]          # 24+ clips x 110 250 samples: 5.3 MB of s16, 10.6 MB of float32
            if k % 2:
                clips = [np.round(c * 32768).astype(np.int16) for c in clips]
            batches.append(clips)
        want = []
        for clips in batches:
            f = [c.astype(np.float32) / np.float32(32768) if c.dtype == np.int16 else c for c in clips]
            r = ref.extract(clips=f)
            want.append([r.clip_hashes(i).copy() for i in range(len(clips))])
        pend, got = [], {}
        for k, clips in enumerate(batches):
            c = ctxs[k % 3]
            c.set_params()
            if len(pend) == 3:
                c0, k0, n0 = pend.pop(0)
                r = c0.fetch(n0, True, False)
                got[k0] = [r.clip_hashes(i).copy() for i in range(n0)]
            pcm = np.concatenate(clips)
            off = np.zeros(len(clips) + 1, np.int64)
            np.cumsum([len(x) for x in clips], out=off[1:])
            assert pcm.nbytes >= 4 << 20
            c.submit(pcm, off)
            pend.append((c, k, len(clips)))
        for c0, k0, n0 in pend:
            r = c0.fetch(n0, True, False)
            got[k0] = [r.clip_hashes(i).copy() for i in range(n0)]
        for k in range(len(batches)):
            for i in range(len(batches[k])):
                assert np.array_equal(got[k][i], want[k][i]), (k, i)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_gpu_large_table_round_trip_through_the_download_ring():
    """afp_table_download of a table of 32 MB or more goes through the pinned ring and the host copy threads
    (download_pageable); 2^17 buckets x 80 = 42 MB, an odd tail chunk included."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    rng = np.random.RandomState(5)
    ht = O.OracleHashTable(hashbits=17, depth=80)
    ht.table[:] = rng.randint(0, 2 ** 32, size=ht.table.shape, dtype=np.uint64).astype(np.uint32)
    ht.counts[:] = rng.randint(1, 200, size=ht.counts.shape).astype(np.int32)
    keep_t, keep_c = ht.table.copy(), ht.counts.copy()
    tb = TableBuilder(ht, Extractor.get(0))                      # non-empty table: uploaded
    ht.table = np.zeros_like(keep_t)
    ht.counts = np.zeros_like(keep_c)
    for _ in range(2):                                            # (second call: the ring and its events are re-used)
        tb.finalize()
        assert np.array_equal(ht.table, keep_t) and np.array_equal(ht.counts, keep_c)
        ht.table = np.zeros_like(keep_t)
        ht.counts = np.zeros_like(keep_c)


def test_afp_error_tells_a_refusal_from_a_failure():
    """only a call the library turned down on its arguments / state (nothing ran) may be retried."""
    from audfprint_amd import _lib
    e = _lib.AfpError('x')
    assert e.status == 0 and not e.refused
    for st, refused in ((-1, True), (-2, True), (-5, True), (-3, False), (-4, False), (-6, False)):
        with pytest.raises(_lib.AfpError) as ei:
            _lib.check(st, 'probe')
        assert ei.value.status == st and ei.value.refused == refused


@pytest.mark.gpu
def test_gpu_refused_store_batch_leaves_no_phantom_ids():
    """store_batch validates BEFORE it files the names (HashTable.name_to_id appends, hash_table.py:340-341): a
    batch that is refused leaves names / hashesperid / the device table as they were; so does a merge the library refuses."""
    from audfprint_amd import _lib
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    z, names = _gold()
    off = z['offsets']
    ht = O.OracleHashTable(hashbits=10, depth=4)
    tb = TableBuilder(ht, Extractor.get(0))
    rows = z['rows'][off[0]:off[2]]
    good = off[0:3] - off[0]
    for bad in (good[:2],                                   # one offset short
                np.array([-1, good[1], good[2]]),           # negative start (the C side would read before the rows)
                np.array([0, good[2] + 5, good[2]]),        # decreasing
                np.array([0, good[1], len(rows) + 1])):     # beyond the rows
        with pytest.raises(ValueError):
            tb.store_batch(['new_a', 'new_b'], rows=rows, offsets=bad)
        assert ht.names == [] and len(ht.hashesperid) == 0
    # device-resident rows that do not belong to these names: a ValueError when the context holds another batch, the
    # library's own refusal (call order) when it holds none -- either way before a name is filed
    fresh = Extractor(0)
    try:
        fresh.set_params()
        tf = TableBuilder(O.OracleHashTable(hashbits=10, depth=4), fresh)
        with pytest.raises(_lib.AfpError) as ei:
            tf.store_batch(['new_a'], offsets=np.array([0, 3]))
        assert ei.value.refused and tf.ht.names == []
    finally:
        fresh.close()
    tb.ex.set_params()
    tb.ex.extract(clips=[O.synth_noise(9100, 2.0), O.synth_noise(9101, 1.0)], want_hashes=True)
    with pytest.raises(ValueError):
        tb.store_batch(['new_a'], offsets=np.array([0, 3]))
    assert ht.names == [] and len(ht.hashesperid) == 0
    # the C entry refuses the negative start by itself
    I32, I64 = __import__('ctypes').POINTER(__import__('ctypes').c_int32), __import__('ctypes').POINTER(__import__('ctypes').c_int64)
    r32 = np.ascontiguousarray(rows, np.int32)
    o64 = np.array([-1, 2, 4], np.int64)
    ids = np.zeros(2, np.int32)
    nov = __import__('ctypes').c_int64()
    assert tb.lib.afp_table_store(tb.ex.h, r32.ctypes.data_as(I32), o64.ctypes.data_as(I64), ids.ctypes.data_as(I32), 2,
                                  __import__('ctypes').byref(nov)) == -1
    # a merge the library refuses (no device buffers behind the pointers) is marked as not committed: names stay
    class _Other(object):
        names, hashesperid, depth, maxtimebits = ['o1'], np.zeros(1, np.uint32), 4, ht.maxtimebits
    with pytest.raises(_lib.AfpError) as ei:
        tb.merge(_Other(), other_device_ptrs=(0, 0))
    assert ei.value.refused and not tb._merge_committed and ht.names == []
    # and the builder still works
    random.seed(1234)
    tb.store_batch(names[:2], rows=rows, offsets=good)
    tb.finalize()
    assert ht.names == names[:2] and int(ht.counts.sum()) > 0


@pytest.mark.gpu
def test_gpu_table_times_beyond_maxtime_wrap_like_the_reference():
    """HashTable.store keeps `time & (maxtime - 1)` (hash_table.py:108-110): a file longer than 16384 frames (6.3 minutes)
    wraps its frame times.  A 500 s clip (21 533 frames) stored from the rows in HBM, next to two short ones: table, counts and
    the hits of a query cut from the part BEYOND the wrap equal the oracle's."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    ex = Extractor.get(0)
    ex.set_params()
    clips = [O.synth_noise(8801, 500.0), O.synth_noise(8802, 7.0), O.synth_noise(8803, 3.0)]
    names = ['long', 'b', 'c']
    r = ex.extract(clips=clips, want_hashes=True)
    assert int(r.clip_hashes(0)[:, 0].max()) > 16384
    ht = O.OracleHashTable(hashbits=20, depth=100)
    tb = TableBuilder(ht, ex)
    random.seed(11)
    tb.store_batch(names, offsets=r.hash_offsets)
    ref = O.OracleHashTable(hashbits=20, depth=100)
    rr = random.Random(11)
    for i, nm in enumerate(names):
        ref.store(nm, r.clip_hashes(i), rr)
    q = ex.extract(clips=[clips[0][11025 * 420:11025 * 428]], want_hashes=True).clip_hashes(0)
    hits = tb.get_hits(q)
    tb.finalize()
    assert np.array_equal(ht.table, ref.table) and np.array_equal(ht.counts, ref.counts)
    assert np.array_equal(np.asarray(ht.hashesperid, np.int64), np.asarray(ref.hashesperid, np.int64))
    assert np.array_equal(hits, ref.get_hits(q)) and len(hits) > 50
