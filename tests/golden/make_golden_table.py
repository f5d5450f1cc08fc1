#!/usr/bin/env python
"""Golden fixture for the hash-table build FROM THE LIVE REFERENCE:
hash_table.HashTable.store called clip by clip with Python's `random` seeded, on a small table
(hashbits 10, depth 4) so that buckets overflow and the RNG path is exercised, plus a default-size
table without overflow.  Run in the build container:  python tests/golden/make_golden_table.py"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import hash_table as RHT  # noqa: E402  (the reference, unchanged)
from oracle import afp_oracle as O  # noqa: E402


def clips_and_hashes(n, secs, seed0):
    out = []
    for i in range(n):
        _, h = O.extract(O.synth_noise(seed0 + i, secs))
        out.append(h)
    return out


def build(hashbits, depth, hashes, names, seed):
    random.seed(seed)
    ht = RHT.HashTable(hashbits=hashbits, depth=depth, maxtime=16384)
    for nm, h in zip(names, hashes):
        ht.store(nm, h)
    return ht


def main():
    hashes = clips_and_hashes(6, 4.0, 9000)
    names = ['clip%d.wav' % i for i in range(6)]
    names[4] = names[1]                       # the same name twice -> same id (name_to_id, hash_table.py:325-344)
    small = build(10, 4, hashes, names, 1234)
    big = build(20, 100, hashes, names, 1234)
    nz = np.nonzero(big.counts)[0]
    # query rows for HashTable.get_hits (hash_table.py:150-176): one stored clip with shifted times + misses
    rng = np.random.RandomState(77)
    q = np.concatenate([hashes[2] + np.array([3, 0], np.int32),
                        np.stack([rng.randint(0, 200, 300), rng.randint(0, 1 << 20, 300)], axis=1).astype(np.int32)])
    small_hits = small.get_hits(q)
    big_hits = big.get_hits(q)
    np.savez_compressed(os.path.join(HERE, 'table_store.npz'),
                        offsets=np.cumsum([0] + [len(h) for h in hashes]).astype(np.int64),
                        rows=np.concatenate(hashes).astype(np.int32),
                        names=np.array(names),
                        small_table=small.table, small_counts=small.counts, small_hpi=small.hashesperid,
                        small_names=np.array(small.names),
                        big_buckets=nz.astype(np.int32), big_rows=big.table[nz], big_counts=big.counts[nz],
                        big_hpi=big.hashesperid, q_rows=q, small_hits=small_hits, big_hits=big_hits)
    print('small: overflowed buckets', int(np.sum(small.counts > 4)), 'total', int(small.counts.sum()))
    print('big: buckets', len(nz), 'max count', int(big.counts.max()))


if __name__ == '__main__':
    main()
