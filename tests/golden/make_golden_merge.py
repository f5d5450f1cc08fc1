#!/usr/bin/env python
"""Golden fixture for HashTable.merge (hash_table.py:291-323) FROM THE LIVE REFERENCE:
two tables built with the reference's store() from disjoint clip sets that share many buckets, merged with
np.random seeded -- on a small table (hashbits 10, depth 4: most shared buckets overflow and take the
np.random.permutation path, some buckets of either table are already over-full before the merge), on the same
data with a deeper receiver (depth 12 <- depth 4: other.depth < self.depth, hardly any overflow) and on the
default-size table (no overflow).  Run in the build container:  python tests/golden/make_golden_merge.py"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import hash_table as RHT  # noqa: E402  (the reference, unchanged)
from oracle import afp_oracle as O  # noqa: E402


def build(hashbits, depth, hashes, names, seed):
    random.seed(seed)
    ht = RHT.HashTable(hashbits=hashbits, depth=depth, maxtime=16384)
    for nm, h in zip(names, hashes):
        ht.store(nm, h)
    return ht


def main():
    hashes = [O.extract(O.synth_noise(9100 + i, 3.0))[1] for i in range(7)]
    names = ['m%d.wav' % i for i in range(7)]
    ha, na, hb, nb = hashes[:4], names[:4], hashes[4:], names[4:]
    out = dict(rows=np.concatenate(hashes).astype(np.int32),
               offsets=np.cumsum([0] + [len(h) for h in hashes]).astype(np.int64), names=np.array(names), nsplit=4)
    for tag, hbits, da, db in (('s', 10, 4, 4), ('d', 10, 12, 4), ('b', 20, 100, 100)):
        a = build(hbits, da, ha, na, 11)
        b = build(hbits, db, hb, nb, 12)
        pre_a_table, pre_a_counts = a.table.copy(), a.counts.copy()
        np.random.seed(4321)
        a.merge(b)
        if hbits <= 12:
            out.update({tag + '_a_table': pre_a_table, tag + '_a_counts': pre_a_counts, tag + '_b_table': b.table,
                        tag + '_b_counts': b.counts, tag + '_m_table': a.table, tag + '_m_counts': a.counts})
        else:                                   # default-size tables: the non-empty buckets only
            nz = np.nonzero(a.counts)[0]
            out.update({tag + '_m_buckets': nz.astype(np.int32), tag + '_m_rows': a.table[nz], tag + '_m_counts': a.counts[nz]})
        out.update({tag + '_m_hpi': a.hashesperid, tag + '_m_names': np.array(a.names),
                    tag + '_a_hpi': np.asarray(build(hbits, da, ha, na, 11).hashesperid),
                    tag + '_b_hpi': b.hashesperid})
        nover = int(np.sum((np.minimum(pre_a_counts, da) + np.minimum(b.counts, db) > da) & (b.counts > 0)))
        print(tag, 'buckets in b', int(np.count_nonzero(b.counts)), 'over-full at merge', nover,
              'a over depth before', int(np.sum(pre_a_counts > da)), 'b over depth', int(np.sum(b.counts > db)))
    np.savez_compressed(os.path.join(HERE, 'table_merge.npz'), **out)


if __name__ == '__main__':
    main()
