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
