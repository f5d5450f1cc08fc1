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
