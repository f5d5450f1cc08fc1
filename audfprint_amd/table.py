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
import os
import random
import time

import numpy as np

import weakref

from . import _lib

# HashTable objects whose host arrays lag their device copy (stores / merges not yet finalized).  Kept here, not as an
# attribute on the user's object: that object is pickled by HashTable.save (hash_table.py:178-190).
_host_stale = weakref.WeakSet()


_native_randint = None


def native_randint_ok(lib):
    """True if afp_mt_randint_replay reproduces THIS interpreter's random.randint(0, n) stream (CPython's
    _randbelow_with_getrandbits over MT19937): checked once per process on a private generator -- 4096 draws across every
    bit length, through a state regeneration -- without touching the global one.  AFP_PY_RANDINT=1 forces the Python loop."""
    global _native_randint
    if _native_randint is None:
        ok = False
        try:
            if not os.environ.get('AFP_PY_RANDINT'):
                g = random.Random(0x5eed)
                st = g.getstate()
                if st[0] == 3 and len(st[1]) == 625:
                    counts = np.array([(1 << (i % 31)) + (i * 2654435761 % (1 << (i % 31))) - (i & 1) for i in range(4096)], dtype=np.int32)
                    counts = np.maximum(counts, 0)
                    want = [g.randint(0, int(c)) for c in counts]
                    mt = np.array(st[1][:624], dtype=np.uint32)
                    pos = C.c_int32(st[1][624])
                    out = np.empty(len(counts), dtype=np.int32)
                    r = lib.afp_mt_randint_replay(mt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  len(counts), out.ctypes.data_as(C.POINTER(C.c_int32)))
                    end = g.getstate()[1]
                    ok = (r == 0 and out.tolist() == want and tuple(mt.tolist()) == tuple(end[:624]) and int(pos.value) == end[624])
        except Exception:
            ok = False
        _native_randint = ok
    return _native_randint


class TableBuilder(object):
    def __init__(self, hashtable, extractor, prefault=False):
        self.ht = hashtable
        self.ex = extractor
        self.lib = extractor.lib
        # host-side seconds spent since creation (perf_counter): the device store (kernels + the wait for them), the overflow
        # replay (fetch, draws, patch), the download -- what a job's stage breakdown reports
        self.seconds = dict(store=0.0, replay=0.0, download=0.0)
        self.overflow_events = 0
        self._index = None                          # name -> id of ht.names, kept between batches (see _ids)
        self._index_list, self._index_len, self._index_last = None, 0, None
        _lib.check(self.lib.afp_table_create(extractor.h, int(hashtable.hashbits), int(hashtable.depth),
                                             int(hashtable.maxtimebits)), 'afp_table_create')
        # The host arrays and the device table start IN STEP: a populated table is uploaded whole; an empty one
        # (HashTable.__init__ / reset: zeros, hash_table.py:61-83) meets the zeroed device table -- unless somebody left
        # values in an "empty" table's rows, in which case it is uploaded too.  While they are in step finalize() moves only
        # the filled prefixes of the rows (afp_table_download_filled).  AFP_TABLE_DENSE_DOWNLOAD=1: always the whole table.
        self.bytes_downloaded = 0
        self._in_step = not os.environ.get('AFP_TABLE_DENSE_DOWNLOAD')
        # "Empty" = every bucket count is zero.  Slots at or beyond a bucket's count are never read by the reference (get_hits
        # takes table[hash, :min(depth, count)], hash_table.py:150-176; merge the same, :304-305) and store() writes only the
        # slot it fills (:117-131), so whatever such slots hold on the host stays there exactly as it would in the reference --
        # no scan of the 420 MB rows (np.any over an empty table touched every page of it).  AFP_TABLE_CHECK_ROWS=1
        # brings the scan back (a non-zero row of an "empty" table is then uploaded like a populated one).
        if int(np.count_nonzero(hashtable.counts)) or (os.environ.get('AFP_TABLE_CHECK_ROWS') and bool(np.any(hashtable.table))):
            table = np.ascontiguousarray(hashtable.table, dtype=np.uint32)
            counts = np.ascontiguousarray(hashtable.counts, dtype=np.int32)
            _lib.check(self.lib.afp_table_upload(extractor.h, table.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 counts.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_upload')
        elif prefault and self._in_step and hashtable.table.nbytes >= (64 << 20) and hashtable.table.flags.c_contiguous:
            # prefault=True (a pipelined job whose host thread is about to wait for its first batches): a fresh table is untouched
            # zero pages -- fault them in behind the scenes NOW, so that finalize() scatters into resident memory (afp_host_prefault:
            # contents unchanged, best effort; 5 ms of eight threads, during which other runtime calls of the process are slow --
            # hence not the default)
            self.lib.afp_host_prefault(C.c_void_p(hashtable.table.ctypes.data), hashtable.table.nbytes)
        self._step_table = hashtable.table            # the array the device is in step with (identity checked at finalize)

    def _ids(self, names):
        """HashTable.store: id_ = self.name_to_id(name, add_if_missing=True) (hash_table.py:95) for every clip of the batch.
        name_to_id searches the `names` list twice per call (:330, :343) -- quadratic over a 12 500-file job -- so the batch
        is resolved through one dict of the current names; new names are appended exactly as name_to_id appends them
        (:340-341, hashesperid grown by np.append with a Python 0).  A table with freed slots (`None` entries left by
        HashTable.remove, which name_to_id re-uses first, :335-338) takes the reference's own method, name by name."""
        ht = self.ht
        cur = ht.names
        if not all(isinstance(n, str) for n in names) or None in cur:
            self._index = None
            return np.array([ht.name_to_id(n, add_if_missing=True) for n in names], dtype=np.int32)
        # the dict is kept between batches while the list is provably the one it was built from (same object, same length,
        # same last entry, no freed slot): a 12 500-file job otherwise rebuilds it from all earlier names in every batch
        index = self._index
        if not (index is not None and self._index_list is cur and self._index_len == len(cur) and
                (not cur or cur[-1] is self._index_last)):
            index = {}
            for i, n in enumerate(cur):
                index.setdefault(n, i)                 # list.index: the first occurrence
        ids = np.empty(len(names), dtype=np.int32)
        nnew = 0
        for k, n in enumerate(names):
            i = index.get(n)
            if i is None:
                i = len(cur)
                cur.append(n)
                index[n] = i
                nnew += 1
            ids[k] = i
        if nnew:
            ht.hashesperid = np.append(ht.hashesperid, [0] * nnew)
        self._index, self._index_list, self._index_len = index, cur, len(cur)
        self._index_last = cur[-1] if cur else None
        return ids

    def store_batch(self, names, rows=None, offsets=None, src=None):
        """Equivalent to ``for name, h in zip(names, per_clip_rows): hashtable.store(name, h)``.
        rows=None: the (time, hash) rows of the LAST extract are used straight from HBM -- of this builder's own extractor,
        or of `src` (another Extractor context on the same GPU whose batch has been waited for: several contexts, uploads
        and kernels overlapping, feed one table in the caller's clip order); `offsets` (rows per clip, CSR; host array) is
        still needed for the per-id hash counts.  Else rows is (N,2) int32 and offsets (nclips+1).
        Returns the number of insertions that met a full bucket (each drew random.randint once, as in the reference)."""
        nclips = len(names)
        I32, I64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        novf = C.c_int64()
        # Everything is checked BEFORE anything is touched -- the device table and the HashTable's own books: _ids()
        # appends new names and grows hashesperid, so a refused batch must be refused first (no phantom ids).
        if offsets is None:
            raise ValueError('offsets (rows per clip, CSR) are needed for the per-id hash counts')
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if rows is None:
            # the rows are attributed to ids by position
            owner = src if src is not None else self.ex
            th, _, _ = owner.counts()                         # waits for the owner's batch
            if len(offsets) != nclips + 1 or int(offsets[0]) != 0 or int(offsets[-1]) != int(th) or owner.last_nclips != nclips:
                raise ValueError('store_batch: %d names / offsets ending at %s do not describe the last batch of the source context '
                                 '(%d clips, %d rows)' % (nclips, offsets[-1] if len(offsets) else None, owner.last_nclips, th))
            if src is not None and src is not self.ex and src.device != self.ex.device:
                raise ValueError('store_batch: src must be a context on the same GPU as the table')
        else:
            rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 2)
            if len(offsets) != nclips + 1 or int(offsets[0]) < 0 or int(offsets[-1]) > len(rows) or np.any(np.diff(offsets) < 0):
                raise ValueError('store_batch: offsets must be %d non-decreasing row offsets into rows' % (nclips + 1))
        ids = self._ids(names)
        t0 = time.perf_counter()
        if rows is None:
            if src is not None and src is not self.ex:
                dh, dho = C.c_void_p(), C.c_void_p()
                _lib.check(self.lib.afp_result_device_ptrs(src.h, C.byref(dh), C.byref(dho), None, None), 'afp_result_device_ptrs')
                _lib.check(self.lib.afp_table_store_device(self.ex.h, dh, dho, th, ids.ctypes.data_as(I32), nclips, C.byref(novf)),
                           'afp_table_store_device')
            else:
                _lib.check(self.lib.afp_table_store(self.ex.h, None, None, ids.ctypes.data_as(I32), nclips, C.byref(novf)),
                           'afp_table_store')
        else:
            _lib.check(self.lib.afp_table_store(self.ex.h, rows.ctypes.data_as(I32), offsets.ctypes.data_as(I64),
                                                ids.ctypes.data_as(I32), nclips, C.byref(novf)), 'afp_table_store')
        # self.hashesperid[id_] += len(timehashpairs)   (hash_table.py:136)
        np.add.at(self.ht.hashesperid, ids, np.diff(offsets).astype(self.ht.hashesperid.dtype))
        t1 = time.perf_counter()
        if novf.value:
            self._replay_overflow(int(novf.value))
        self.seconds['store'] += t1 - t0
        self.seconds['replay'] += time.perf_counter() - t1
        self.overflow_events += int(novf.value)
        self.ht.dirty = True
        _host_stale.add(self.ht)                   # the device table is ahead of ht.table / ht.counts until finalize()
        return int(novf.value)

    def _replay_overflow(self, novf):
        """One random.randint(0, count) per insertion into a full bucket, in insertion order, on Python's GLOBAL generator
        (hash_table.py:128) -- drawn by the library from the generator's own state (afp_table_replay_overflow: CPython's
        algorithm over the same Mersenne-Twister words; the state is put back afterwards, so the process's random stream
        continues exactly as if Python had made the calls).  If the library's stream ever differed from this interpreter's
        (checked once per process), the draws are made by Python itself."""
        if native_randint_ok(self.lib):
            st = random.getstate()
            mt = np.array(st[1][:624], dtype=np.uint32)
            pos = C.c_int32(st[1][624])
            nw = C.c_int64()
            _lib.check(self.lib.afp_table_replay_overflow(self.ex.h, mt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), C.byref(nw)),
                       'afp_table_replay_overflow')
            random.setstate((st[0], tuple(mt.tolist()) + (int(pos.value),), st[2]))
            return
        I32 = C.POINTER(C.c_int32)
        ev = np.empty((novf, 4), dtype=np.int32)
        _lib.check(self.lib.afp_table_fetch_overflow(self.ex.h, ev.ctypes.data_as(I32)), 'afp_table_fetch_overflow')
        ev = ev[np.argsort(ev[:, 0].astype(np.uint32), kind='stable')]      # the reference's insertion order
        depth = int(self.ht.depth)
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

    def merge(self, other, other_device_ptrs=None, packed=False):
        """``hashtable.merge(other)`` (hash_table.py:291-323) with the bucket work on the device table.
        ``other`` is a reference-style HashTable (host arrays are uploaded), or -- with
        ``other_device_ptrs=(table_ptr, counts_ptr)`` -- anything carrying ``names, hashesperid, depth,
        maxtimebits`` whose table already sits in HBM (another GPU's table received over xGMI).
        Over-full buckets draw ``np.random.permutation`` on the host in the reference's bucket order, so with
        the same ``np.random.seed`` the merged table is bit-identical.  Returns the number of such buckets.
        ``packed=True``: the other table comes in its PACKED form (``TableBuilder.pack`` on the sending side: counts + the
        filled prefixes of its rows) -- ``other.table`` is then the flat value array, ``other_device_ptrs`` =
        (values_ptr, counts_ptr)."""
        ht = self.ht
        assert ht.maxtimebits == other.maxtimebits                              # :295
        self._merge_committed = False
        ncurrent = len(ht.names)                                                # :296
        odepth = int(other.depth)
        nov = C.c_int64()
        if other_device_ptrs is not None:
            fn = self.lib.afp_table_merge_packed_device if packed else self.lib.afp_table_merge_device
            _lib.check(fn(self.ex.h, C.c_void_p(int(other_device_ptrs[0])), C.c_void_p(int(other_device_ptrs[1])), odepth, ncurrent,
                          C.byref(nov)), 'afp_table_merge_packed_device' if packed else 'afp_table_merge_device')
        elif packed:
            vals = np.ascontiguousarray(other.table, dtype=np.uint32).reshape(-1)
            ocnt = np.ascontiguousarray(other.counts, dtype=np.int32)
            if ocnt.shape[0] != (1 << int(ht.hashbits)):
                raise ValueError('merge needs tables with the same hashbits')
            _lib.check(self.lib.afp_table_merge_packed(self.ex.h, vals.ctypes.data_as(C.POINTER(C.c_uint32)), vals.shape[0],
                                                       ocnt.ctypes.data_as(C.POINTER(C.c_int32)), odepth, ncurrent, C.byref(nov)),
                       'afp_table_merge_packed')
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
        # The bookkeeping follows the device call: a call the library REFUSES (AfpError.refused: bad argument / parameter /
        # state, decided before anything ran) leaves the table, names and hashesperid as they were, and only such a call may
        # be retried with other arguments (shard.merge_tables_to_rank0).  From here on the merge has happened on the device:
        # an error below must not be answered by merging the same table again.
        self._merge_committed = True
        ht.names += other.names                                                 # :298
        ht.hashesperid = np.append(ht.hashesperid, other.hashesperid)           # :299
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

    def pack(self):
        """Build the packed form of the device table in HBM (include/afp.h: counts + the filled prefix of every row, bucket
        after bucket): what a rank ships to the merging rank instead of the whole table.  Returns the number of values."""
        tot = C.c_int64()
        _lib.check(self.lib.afp_table_pack(self.ex.h, C.byref(tot)), 'afp_table_pack')
        return int(tot.value)

    def packed_device_ptrs(self):
        """(values_ptr, counts_ptr, n_values) of the last pack()."""
        v, c, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        _lib.check(self.lib.afp_table_packed_device_ptrs(self.ex.h, C.byref(v), C.byref(c), C.byref(n)), 'afp_table_packed_device_ptrs')
        return v.value, c.value, int(n.value)

    def fetch_packed(self):
        """Host copies (values uint32[n], counts int32[2^hashbits]) of the last pack()."""
        _, _, n = self.packed_device_ptrs()
        vals = np.empty(n, dtype=np.uint32)
        counts = np.empty(1 << int(self.ht.hashbits), dtype=np.int32)
        _lib.check(self.lib.afp_table_fetch_packed(self.ex.h, vals.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                   counts.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_fetch_packed')
        return vals, counts

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
        t0 = time.perf_counter()
        table = np.ascontiguousarray(ht.table, dtype=np.uint32)
        counts = np.ascontiguousarray(ht.counts, dtype=np.int32)
        # in step = ht.table is still the very array the device table was created / uploaded / last downloaded against
        # (no copy was needed to make it contiguous uint32): then only the filled prefixes move
        if self._in_step and table is ht.table and ht.table is self._step_table:
            n = C.c_int64()
            _lib.check(self.lib.afp_table_download_filled(self.ex.h, table.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                          counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)), 'afp_table_download_filled')
            self.bytes_downloaded += 4 * int(n.value) + counts.nbytes
        else:
            _lib.check(self.lib.afp_table_download(self.ex.h, table.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                   counts.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_download')
            self.bytes_downloaded += table.nbytes + counts.nbytes
        ht.table = table
        ht.counts = counts
        self._step_table = table
        ht.dirty = True
        _host_stale.discard(ht)
        self.seconds['download'] += time.perf_counter() - t0
        return ht
