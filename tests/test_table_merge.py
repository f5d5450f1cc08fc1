"""SURVEY.md §8f row f1, second half: HashTable.merge (hash_table.py:291-323).  CPU: the oracle's restatement
vs the golden made by the live reference (tests/golden/make_golden_merge.py); GPU: TableBuilder.merge over the
device-resident table vs the same golden, bit-exact including the np.random.permutation path of over-full
buckets -- and the store -> finalize -> store -> finalize sequence whose overflow writes must survive."""
import os
import random

import numpy as np
import pytest

from oracle import afp_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'table_merge.npz')
CASES = (('s', 10, 4, 4), ('d', 10, 12, 4))


def _gold():
    z = np.load(GOLD)
    names = [str(n) for n in z['names']]
    return z, names, int(z['nsplit'])


def _oracle_pair(z, names, nsplit, hbits, da, db):
    off = z['offsets']
    a = O.OracleHashTable(hashbits=hbits, depth=da)
    b = O.OracleHashTable(hashbits=hbits, depth=db)
    ra, rb = random.Random(11), random.Random(12)
    for i, nm in enumerate(names):
        (a if i < nsplit else b).store(nm, z['rows'][off[i]:off[i + 1]], ra if i < nsplit else rb)
    return a, b


@pytest.mark.parametrize('case', CASES)
def test_oracle_merge_equals_reference(case):
    tag, hbits, da, db = case
    z, names, nsplit = _gold()
    a, b = _oracle_pair(z, names, nsplit, hbits, da, db)
    assert np.array_equal(a.table, z[tag + '_a_table']) and np.array_equal(b.table, z[tag + '_b_table'])
    a.merge(b, np.random.RandomState(4321))
    assert np.array_equal(a.counts, z[tag + '_m_counts'])
    assert np.array_equal(a.table, z[tag + '_m_table'])
    assert np.array_equal(a.hashesperid, z[tag + '_m_hpi']) and a.names == [str(n) for n in z[tag + '_m_names']]


def test_oracle_merge_default_size():
    z, names, nsplit = _gold()
    a, b = _oracle_pair(z, names, nsplit, 20, 100, 100)
    a.merge(b, np.random.RandomState(4321))
    nz = z['b_m_buckets']
    assert np.array_equal(np.nonzero(a.counts)[0], nz)
    assert np.array_equal(a.table[nz], z['b_m_rows']) and np.array_equal(a.counts[nz], z['b_m_counts'])


@pytest.mark.gpu
@pytest.mark.parametrize('device_side', [False, True])
@pytest.mark.parametrize('case', CASES)
def test_gpu_merge_equals_reference(case, device_side):
    """The receiving table is built ON THE GPU (store_batch incl. overflow replay), the other table arrives as
    host arrays or as device pointers of a second builder on the same GPU."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    tag, hbits, da, db = case
    z, names, nsplit = _gold()
    off = z['offsets']
    ex = Extractor.get(0)
    a = O.OracleHashTable(hashbits=hbits, depth=da)
    tb = TableBuilder(a, ex)
    random.seed(11)
    tb.store_batch(names[:nsplit], rows=z['rows'][:off[nsplit]], offsets=off[:nsplit + 1])
    np.random.seed(4321)
    if device_side:
        ex2 = Extractor(0)
        b = O.OracleHashTable(hashbits=hbits, depth=db)
        tb2 = TableBuilder(b, ex2)
        random.seed(12)
        tb2.store_batch(names[nsplit:], rows=z['rows'][off[nsplit]:], offsets=off[nsplit:] - off[nsplit])
        nov = tb.merge(b, other_device_ptrs=tb2.device_ptrs())
        ex2.close()
    else:
        _, b = _oracle_pair(z, names, nsplit, hbits, da, db)
        nov = tb.merge(b)
    tb.finalize()
    want_over = int(np.sum((np.minimum(z[tag + '_a_counts'], da) + np.minimum(z[tag + '_b_counts'], db) > da)
                           & (z[tag + '_b_counts'] > 0)))
    assert nov == want_over
    assert np.array_equal(a.counts, z[tag + '_m_counts'])
    assert np.array_equal(a.table, z[tag + '_m_table'])
    assert np.array_equal(a.hashesperid, z[tag + '_m_hpi']) and a.names == [str(n) for n in z[tag + '_m_names']]
    # the merged table is live on the device: a query sees what the host copy says
    q = z['rows'][off[nsplit]:off[nsplit] + 200]
    assert np.array_equal(tb.get_hits(q), a.get_hits(q))


@pytest.mark.gpu
@pytest.mark.parametrize('device_side', [False, True])
@pytest.mark.parametrize('case', CASES)
def test_gpu_merge_from_the_packed_form_equals_reference(case, device_side):
    """the other table arrives PACKED -- counts + the filled prefixes of its rows (TableBuilder.pack on the
    sending builder) -- as host arrays (afp_table_merge_packed) or device pointers (afp_table_merge_packed_device: what
    crosses xGMI).  Same golden as the dense merge, permutation path of over-full buckets included."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.shard import _RemoteTable, pack_host
    from audfprint_amd.table import TableBuilder
    tag, hbits, da, db = case
    z, names, nsplit = _gold()
    off = z['offsets']
    ex = Extractor.get(0)
    a = O.OracleHashTable(hashbits=hbits, depth=da)
    tb = TableBuilder(a, ex)
    random.seed(11)
    tb.store_batch(names[:nsplit], rows=z['rows'][:off[nsplit]], offsets=off[:nsplit + 1])
    ex2 = Extractor(0)
    b = O.OracleHashTable(hashbits=hbits, depth=db)
    tb2 = TableBuilder(b, ex2)
    random.seed(12)
    tb2.store_batch(names[nsplit:], rows=z['rows'][off[nsplit]:], offsets=off[nsplit:] - off[nsplit])
    n = tb2.pack()
    vals, cnts = tb2.fetch_packed()
    tb2.finalize()
    assert n == len(vals) == int(np.minimum(b.counts, db).sum()) and np.array_equal(cnts, b.counts)
    assert np.array_equal(vals, pack_host(b.table, b.counts, db))          # the numpy restatement the CPU tests use
    np.random.seed(4321)
    if device_side:
        vp, cp, n2 = tb2.packed_device_ptrs()
        assert n2 == n
        nov = tb.merge(_RemoteTable(b.names, b.hashesperid, db, b.maxtimebits), other_device_ptrs=(vp, cp), packed=True)
    else:
        nov = tb.merge(_RemoteTable(b.names, b.hashesperid, db, b.maxtimebits, table=vals, counts=cnts), packed=True)
    ex2.close()
    tb.finalize()
    want_over = int(np.sum((np.minimum(z[tag + '_a_counts'], da) + np.minimum(z[tag + '_b_counts'], db) > da)
                           & (z[tag + '_b_counts'] > 0)))
    assert nov == want_over
    assert np.array_equal(a.counts, z[tag + '_m_counts'])
    assert np.array_equal(a.table, z[tag + '_m_table'])
    assert np.array_equal(a.hashesperid, z[tag + '_m_hpi']) and a.names == [str(n) for n in z[tag + '_m_names']]


@pytest.mark.gpu
def test_gpu_merge_packed_refuses_a_stream_that_does_not_match_its_counts():
    from audfprint_amd import _lib
    from audfprint_amd.batch import Extractor
    from audfprint_amd.shard import _RemoteTable
    from audfprint_amd.table import TableBuilder
    a = O.OracleHashTable(hashbits=10, depth=4)
    tb = TableBuilder(a, Extractor.get(0))
    cnts = np.zeros(1024, np.int32)
    cnts[5] = 3
    with pytest.raises(_lib.AfpError):
        tb.merge(_RemoteTable(['x'], np.array([3]), 4, a.maxtimebits, table=np.arange(2, dtype=np.uint32), counts=cnts), packed=True)
    assert a.names == [] and len(a.hashesperid) == 0                       # a refused merge leaves the books alone
    tb.finalize()
    assert not a.counts.any()


@pytest.mark.gpu
def test_gpu_sparse_download_equals_dense_download_on_a_full_size_table():
    """finalize() moves counts + filled prefixes through the pinned ring and scatters them into the host rows
    (afp_table_download_filled); the table here is the default 2^20 x 100 with ~9.5 entries per bucket (40 MB packed: five
    ring chunks, every scatter thread crosses bucket boundaries) and a few thousand over-full buckets.  Against the dense
    download of the same device table, and a second store + finalize on top (the host array stays in step)."""
    import ctypes as C
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    rng = np.random.RandomState(77)
    ex = Extractor.get(0)
    ht = O.OracleHashTable(hashbits=20, depth=100)
    tb = TableBuilder(ht, ex)
    assert ex.lib.afp_host_threads() >= 1
    for rep in range(2):
        nrows = 5000000
        rows = np.empty((nrows, 2), np.int32)
        rows[:, 0] = rng.randint(0, 16384, size=nrows)
        h = rng.randint(0, 1 << 20, size=nrows)
        h[:400000] = rng.randint(0, 2000, size=400000)               # 2000 popular hashes: ~200 entries each, far over the depth
        rows[:, 1] = h
        offsets = np.linspace(0, nrows, 101).astype(np.int64)
        random.seed(3 + rep)
        tb.store_batch(['r%dc%d' % (rep, i) for i in range(100)], rows=rows, offsets=offsets)
        before = tb.bytes_downloaded
        tb.finalize()
        moved = tb.bytes_downloaded - before
        dense_t = np.zeros_like(ht.table)
        dense_c = np.zeros_like(ht.counts)
        _lib_check = __import__('audfprint_amd._lib', fromlist=['check']).check
        _lib_check(ex.lib.afp_table_download(ex.h, dense_t.ctypes.data_as(C.POINTER(C.c_uint32)), dense_c.ctypes.data_as(C.POINTER(C.c_int32))))
        assert np.array_equal(ht.counts, dense_c) and int(np.sum(dense_c > 100)) > 1000
        assert np.array_equal(ht.table, dense_t)
        assert moved == 4 * int(np.minimum(dense_c, 100).sum()) + dense_c.nbytes and moved < dense_t.nbytes // 4


@pytest.mark.gpu
def test_gpu_merge_default_size_then_store():
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    z, names, nsplit = _gold()
    off = z['offsets']
    a = O.OracleHashTable(hashbits=20, depth=100)
    tb = TableBuilder(a, Extractor.get(0))
    tb.store_batch(names[:nsplit], rows=z['rows'][:off[nsplit]], offsets=off[:nsplit + 1])
    _, b = _oracle_pair(z, names, nsplit, 20, 100, 100)
    np.random.seed(4321)
    assert tb.merge(b) == 0
    tb.finalize()
    nz = z['b_m_buckets']
    assert np.array_equal(np.nonzero(a.counts)[0], nz)
    assert np.array_equal(a.table[nz], z['b_m_rows']) and np.array_equal(a.counts[nz], z['b_m_counts'])
    assert np.array_equal(a.hashesperid, z['b_m_hpi'])


@pytest.mark.gpu
def test_gpu_store_finalize_store_finalize_keeps_overflow_writes():
    """finalize() used to apply the replayed overflow writes to the host copy only; a later
    store_batch + finalize (or get_hits) then worked on a device table without them."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    z = np.load(os.path.join(os.path.dirname(GOLD), 'table_store.npz'))
    names = [str(n) for n in z['names']]
    off = z['offsets']
    ht = O.OracleHashTable(hashbits=10, depth=4)
    tb = TableBuilder(ht, Extractor.get(0))
    ref = O.OracleHashTable(hashbits=10, depth=4)
    rr = random.Random(1234)
    random.seed(1234)
    for a, b in ((0, 3), (3, 6)):
        tb.store_batch(names[a:b], rows=z['rows'][off[a]:off[b]], offsets=off[a:b + 1] - off[a])
        for i in range(a, b):
            ref.store(names[i], z['rows'][off[i]:off[i + 1]], rr)
        tb.finalize()                                        # user-visible finalize in the middle
        assert np.array_equal(ht.table, ref.table) and np.array_equal(ht.counts, ref.counts)
        assert np.array_equal(tb.get_hits(z['q_rows']), ref.get_hits(z['q_rows']))
    assert np.array_equal(ht.table, z['small_table']) and np.array_equal(ht.counts, z['small_counts'])


@pytest.mark.gpu
def test_merge_refuses_a_table_whose_stores_were_not_finalized():
    """TableBuilder.merge(other) uploads other's HOST arrays; if other is wrapped by a builder that has stored
    since its last finalize(), those arrays are stale -- refuse instead of merging them silently."""
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    ex = Extractor.get(0)
    a = O.OracleHashTable(hashbits=12, depth=8)
    b = O.OracleHashTable(hashbits=12, depth=8)
    rows = np.array([[1, 5], [2, 9], [3, 5]], np.int32)
    tb = TableBuilder(b, ex)
    tb.store_batch(['x'], rows=rows, offsets=np.array([0, 3], np.int64))
    assert int(b.counts.sum()) == 0                 # the host arrays lag the device table
    tb.finalize()
    assert int(b.counts.sum()) == 3
    tb.store_batch(['y'], rows=rows, offsets=np.array([0, 3], np.int64))      # stale again
    ta = TableBuilder(a, ex)
    with pytest.raises(ValueError):
        ta.merge(b)
