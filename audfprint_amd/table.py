"""Batch build of the reference hash table on the GPU ("next" row f1).

`TableBuilder` wraps a reference-style ``HashTable`` object (hash_table.py:50-83: attributes
``table``, ``counts``, ``names``, ``hashesperid``, ``hashbits``, ``depth``, ``maxtimebits`` and the
method ``name_to_id``) and replaces the per-hash Python loop of ``HashTable.store``
(hash_table.py:91-138) for whole batches.  The device table is authoritative between
``store_batch`` calls; ``finalize()`` copies it back into the HashTable's numpy arrays, after
which the object pickles / saves / matches exactly as the reference's would.

Bit-exactness: slots are filled in the reference's insertion order, so without overflow the
arrays are identical.  An insertion into a full bucket draws ``random.randint(0, count)`` in the
reference (:128-131); the GPU reports those insertions as ordered events and this class replays
them with the same calls on Python's global ``random`` -- seed it the same and the tables match.
"""
import ctypes as C
import random

import numpy as np

from . import _lib


class TableBuilder(object):
    def __init__(self, hashtable, extractor):
        self.ht = hashtable
        self.ex = extractor
        self.lib = extractor.lib
        self._patches = []                      # (bucket, slot, value) from replayed overflow insertions, in order
        _lib.check(self.lib.afp_table_create(extractor.h, int(hashtable.hashbits), int(hashtable.depth),
                                             int(hashtable.maxtimebits)), 'afp_table_create')
        if int(np.count_nonzero(hashtable.counts)):
            table = np.ascontiguousarray(hashtable.table, dtype=np.uint32)
            counts = np.ascontiguousarray(hashtable.counts, dtype=np.int32)
            _lib.check(self.lib.afp_table_upload(extractor.h, table.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 counts.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_upload')

    def _ids(self, names):
        # HashTable.store: id_ = self.name_to_id(name, add_if_missing=True)   (hash_table.py:95)
        return np.array([self.ht.name_to_id(n, add_if_missing=True) for n in names], dtype=np.int32)

    def store_batch(self, names, rows=None, offsets=None):
        """Equivalent to ``for name, h in zip(names, per_clip_rows): hashtable.store(name, h)``.
        With rows=None the (time, hash) rows of the extractor's LAST extract are used straight from
        HBM (offsets then come from that result); else rows is (N,2) int32 and offsets (nclips+1)."""
        nclips = len(names)
        ids = self._ids(names)
        I32, I64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        novf = C.c_int64()
        if rows is None:
            if offsets is None:
                raise ValueError('offsets (rows per clip, CSR) are needed for the per-id hash counts')
            offsets = np.ascontiguousarray(offsets, dtype=np.int64)
            _lib.check(self.lib.afp_table_store(self.ex.h, None, None, ids.ctypes.data_as(I32), nclips, C.byref(novf)),
                       'afp_table_store')
        else:
            rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 2)
            offsets = np.ascontiguousarray(offsets, dtype=np.int64)
            _lib.check(self.lib.afp_table_store(self.ex.h, rows.ctypes.data_as(I32), offsets.ctypes.data_as(I64),
                                                ids.ctypes.data_as(I32), nclips, C.byref(novf)), 'afp_table_store')
        # self.hashesperid[id_] += len(timehashpairs)   (hash_table.py:136)
        for i, id_ in enumerate(ids):
            self.ht.hashesperid[id_] += int(offsets[i + 1] - offsets[i])
        if novf.value:
            ev = np.empty((novf.value, 4), dtype=np.int32)
            _lib.check(self.lib.afp_table_fetch_overflow(self.ex.h, ev.ctypes.data_as(I32)), 'afp_table_fetch_overflow')
            ev = ev[np.argsort(ev[:, 0].astype(np.uint32), kind='stable')]      # the reference's insertion order
            depth = int(self.ht.depth)
            for _, bucket, val, count in ev.tolist():
                slot = random.randint(0, count)                                 # hash_table.py:128
                if slot < depth:                                                # :130-131
                    self._patches.append((bucket, slot, np.uint32(val & 0xFFFFFFFF)))
        self.ht.dirty = True
        return int(novf.value)

    def _sync_device(self):
        """Replayed overflow writes live on the host until finalize(); bring the device table up to date."""
        if self._patches:
            ht = self.finalize()
            table = np.ascontiguousarray(ht.table, dtype=np.uint32)
            counts = np.ascontiguousarray(ht.counts, dtype=np.int32)
            _lib.check(self.lib.afp_table_upload(self.ex.h, table.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 counts.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_upload')

    def get_hits(self, hashes):
        """HashTable.get_hits (hash_table.py:150-176) over the device-resident table: (nhits, 4) int32
        rows [id, delta_time, hash, time] for the (N,2) [time, hash] query rows, reference order."""
        self._sync_device()
        rows = np.ascontiguousarray(np.asarray(hashes, dtype=np.int32).reshape(-1, 2))
        nh = C.c_int64()
        I32 = C.POINTER(C.c_int32)
        _lib.check(self.lib.afp_table_get_hits(self.ex.h, rows.ctypes.data_as(I32), rows.shape[0], C.byref(nh)),
                   'afp_table_get_hits')
        hits = np.zeros((nh.value, 4), dtype=np.int32)
        _lib.check(self.lib.afp_table_fetch_hits(self.ex.h, hits.ctypes.data_as(I32)), 'afp_table_fetch_hits')
        return hits

    def finalize(self):
        """Copy the device table into the HashTable object and apply the replayed overflow writes."""
        ht = self.ht
        table = np.ascontiguousarray(ht.table, dtype=np.uint32)
        counts = np.ascontiguousarray(ht.counts, dtype=np.int32)
        _lib.check(self.lib.afp_table_download(self.ex.h, table.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               counts.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_download')
        for bucket, slot, val in self._patches:
            table[bucket, slot] = val
        self._patches = []
        ht.table = table
        ht.counts = counts
        ht.dirty = True
        return ht
