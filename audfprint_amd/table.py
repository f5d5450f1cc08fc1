"""Batch build of the reference hash table on the GPU ("next" row f1).

`TableBuilder` wraps a reference-style ``HashTable`` object (hash_table.py:50-83: attributes
``table``, ``counts``, ``names``, ``hashesperid``, ``hashbits``, ``depth``, ``maxtimebits`` and the
method ``name_to_id``) and replaces the per-hash Python loop of ``HashTable.store``
(hash_table.py:91-138) for whole batches.  ``TableBuilder.merge`` does the same for ``HashTable.merge`` (:291-323).  The device table is the
authoritative copy at all times (host-decided random writes are patched into it at once);
``finalize()`` copies it back into the HashTable's numpy arrays, after which the object
pickles / saves / matches exactly as the reference's would.

Bit-exactness: slots are filled in the reference's insertion order, so without overflow the
arrays are identical.  An insertion into a full bucket draws ``random.randint(0, count)`` in the
reference (:128-131); the GPU reports those insertions as ordered events and this class replays
them with the same calls on Python's global ``random`` -- seed it the same and the tables match.
``merge`` likewise leaves the ``np.random.permutation`` of over-full buckets (:312) to the host.
"""
import ctypes as C
import random

import numpy as np

import weakref

from . import _lib

# HashTable objects whose host arrays lag their device copy (stores / merges not yet finalized).  Kept here, not as an
# attribute on the user's object: that object is pickled by HashTable.save (hash_table.py:178-190).
_host_stale = weakref.WeakSet()


class TableBuilder(object):
    def __init__(self, hashtable, extractor):
        self.ht = hashtable
        self.ex = extractor
        self.lib = extractor.lib
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
        np.add.at(self.ht.hashesperid, ids, np.diff(offsets).astype(self.ht.hashesperid.dtype))
        if novf.value:
            ev = np.empty((novf.value, 4), dtype=np.int32)
            _lib.check(self.lib.afp_table_fetch_overflow(self.ex.h, ev.ctypes.data_as(I32)), 'afp_table_fetch_overflow')
            ev = ev[np.argsort(ev[:, 0].astype(np.uint32), kind='stable')]      # the reference's insertion order
            depth = int(self.ht.depth)
            # one random.randint(0, count) per event, in order (hash_table.py:128) -- the only part that has to be a
            # Python loop; everything around it is vectorised
            rr = random.randint
            slots = np.fromiter((rr(0, c) for c in ev[:, 3].tolist()), dtype=np.int64, count=len(ev))
            keep = np.nonzero(slots < depth)[0]                                 # :130-131
            if len(keep):
                key = ev[keep, 1].astype(np.int64) * depth + slots[keep]
                # later writes to a slot win, as in the loop: keep the LAST event of every (bucket, slot)
                _, first_rev = np.unique(key[::-1], return_index=True)
                last = keep[len(keep) - 1 - first_rev]
                arr = np.empty((len(last), 3), dtype=np.int32)
                arr[:, 0] = ev[last, 1]
                arr[:, 1] = slots[last]
                arr[:, 2] = ev[last, 2]
                arr = np.ascontiguousarray(arr)
                _lib.check(self.lib.afp_table_patch(self.ex.h, arr.ctypes.data_as(C.POINTER(C.c_int32)), arr.shape[0]),
                           'afp_table_patch')
        self.ht.dirty = True
        _host_stale.add(self.ht)                   # the device table is ahead of ht.table / ht.counts until finalize()
        return int(novf.value)

    def merge(self, other, other_device_ptrs=None):
        """``hashtable.merge(other)`` (hash_table.py:291-323) with the bucket work on the device table.
        ``other`` is a reference-style HashTable (host arrays are uploaded), or -- with
        ``other_device_ptrs=(table_ptr, counts_ptr)`` -- anything carrying ``names, hashesperid, depth,
        maxtimebits`` whose table already sits in HBM (another GPU's table received over xGMI).
        Over-full buckets draw ``np.random.permutation`` on the host in the reference's bucket order, so with
        the same ``np.random.seed`` the merged table is bit-identical.  Returns the number of such buckets."""
        ht = self.ht
        assert ht.maxtimebits == other.maxtimebits                              # :295
        ncurrent = len(ht.names)                                                # :296
        ht.names += other.names                                                 # :298
        ht.hashesperid = np.append(ht.hashesperid, other.hashesperid)           # :299
        odepth = int(other.depth)
        nov = C.c_int64()
        if other_device_ptrs is not None:
            _lib.check(self.lib.afp_table_merge_device(self.ex.h, C.c_void_p(int(other_device_ptrs[0])),
                                                       C.c_void_p(int(other_device_ptrs[1])), odepth, ncurrent,
                                                       C.byref(nov)), 'afp_table_merge_device')
        else:
            if other.table.shape[0] != (1 << int(ht.hashbits)):
                raise ValueError('merge needs tables with the same hashbits')
            if other in _host_stale:
                # `other` is wrapped by another TableBuilder whose device copy is ahead of these host arrays
                raise ValueError('merge: the other table has stores that were not finalized (call its TableBuilder.finalize())')
            otab = np.ascontiguousarray(other.table, dtype=np.uint32)
            ocnt = np.ascontiguousarray(other.counts, dtype=np.int32)
            _lib.check(self.lib.afp_table_merge(self.ex.h, otab.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                ocnt.ctypes.data_as(C.POINTER(C.c_int32)), odepth, ncurrent,
                                                C.byref(nov)), 'afp_table_merge')
        n = int(nov.value)
        if n:
            depth = int(ht.depth)
            buckets = np.zeros(n, np.int32)
            nvals = np.zeros(n, np.int32)
            allvals = np.zeros((n, depth + odepth), np.uint32)
            _lib.check(self.lib.afp_table_fetch_merge_overflow(self.ex.h, buckets.ctypes.data_as(C.POINTER(C.c_int32)),
                                                               nvals.ctypes.data_as(C.POINTER(C.c_int32)),
                                                               allvals.ctypes.data_as(C.POINTER(C.c_uint32))),
                       'afp_table_fetch_merge_overflow')
            rows = np.empty((n, depth), np.uint32)
            for i in range(n):
                rows[i] = np.random.permutation(allvals[i, :nvals[i]])[:depth]  # :312
            arr = np.empty((n, depth, 3), np.int32)
            arr[:, :, 0] = buckets[:, None]
            arr[:, :, 1] = np.arange(depth, dtype=np.int32)[None, :]
            arr[:, :, 2] = rows.view(np.int32)
            arr = np.ascontiguousarray(arr.reshape(-1, 3))
            _lib.check(self.lib.afp_table_patch(self.ex.h, arr.ctypes.data_as(C.POINTER(C.c_int32)), arr.shape[0]),
                       'afp_table_patch')
        ht.dirty = True                                                         # :323
        _host_stale.add(ht)
        return n

    def clip_counts(self):
        """counts[k] = min(counts[k], depth) on the device table: what the reference's parent holds after merging this table
        into an EMPTY one (hash_table.py:304-305, 315-321), as `multiproc_add` does with every worker's table -- core 0's
        included (audfprint.py:226-235)."""
        _lib.check(self.lib.afp_table_clip_counts(self.ex.h), 'afp_table_clip_counts')
        self.ht.dirty = True
        _host_stale.add(self.ht)

    def device_ptrs(self):
        """(table_ptr, counts_ptr): device addresses of this builder's table, for merge(..., other_device_ptrs=)."""
        t, c = C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.afp_table_device_ptrs(self.ex.h, C.byref(t), C.byref(c)), 'afp_table_device_ptrs')
        return t.value, c.value

    def _sync_device(self):
        """The device table is always up to date (host-decided writes are patched in at once); kept for callers."""

    def get_hits(self, hashes):
        """HashTable.get_hits (hash_table.py:150-176) over the device-resident table: (nhits, 4) int32
        rows [id, delta_time, hash, time] for the (N,2) [time, hash] query rows, reference order."""
        rows = np.ascontiguousarray(np.asarray(hashes, dtype=np.int32).reshape(-1, 2))
        nh = C.c_int64()
        I32 = C.POINTER(C.c_int32)
        _lib.check(self.lib.afp_table_get_hits(self.ex.h, rows.ctypes.data_as(I32), rows.shape[0], C.byref(nh)),
                   'afp_table_get_hits')
        hits = np.zeros((nh.value, 4), dtype=np.int32)
        _lib.check(self.lib.afp_table_fetch_hits(self.ex.h, hits.ctypes.data_as(I32)), 'afp_table_fetch_hits')
        return hits

    def finalize(self):
        """Copy the device table into the HashTable object (it then pickles / saves / matches as the
        reference's would).  The device copy stays valid: more store_batch / merge / get_hits calls may follow."""
        ht = self.ht
        table = np.ascontiguousarray(ht.table, dtype=np.uint32)
        counts = np.ascontiguousarray(ht.counts, dtype=np.int32)
        _lib.check(self.lib.afp_table_download(self.ex.h, table.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               counts.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_download')
        ht.table = table
        ht.counts = counts
        ht.dirty = True
        _host_stale.discard(ht)
        return ht
