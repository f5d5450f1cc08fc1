"""Query-side vote counting on the GPU ("next" row f4).

``match_hashes(matcher, tb, hashes)`` gives what ``Matcher.match_hashes(ht, hashes)``
(audfprint_match.py:314-352) gives, with the table walk and the histogram work done on the device-resident
table of a ``TableBuilder``:

* ``HashTable.get_hits`` (hash_table.py:150-176)                       -> ``afp_table_get_hits``
* ``np.unique`` / ``np.bincount`` of ``_best_count_ids`` (:128-132)    -> ``afp_table_count_ids``
* the per-id ``np.bincount`` loop of ``_approx_match_counts`` (:289)   -> ``afp_table_skew_hist``

What stays on the host is a few numpy calls over arrays the size of the candidate list (the weighting and
``np.argsort`` of :133-147 -- numpy's own sort decides ties, so it has to be numpy's -- and the mode picking of
:291-311).  ``matcher`` is the caller's ``audfprint_match.Matcher`` (or anything with its attributes
``window, threshcount, search_depth, max_alignments_per_id, exact_count, find_time_range``); the options
that need the hit rows themselves (``exact_count``, ``find_time_range``, ``hashesfor``) download the hits and
run the matcher's own methods on them.
"""
import ctypes as C

import numpy as np

from . import _lib
from .audfprint_analyze import locmax


class VoteCounter(object):
    """Histogram side of the matcher over the hits of ``tb.get_hits`` that are still in HBM."""

    def __init__(self, tb):
        self.tb = tb
        self.lib = tb.lib
        self.h = tb.ex.h

    def query(self, hashes):
        """Walk the table for the (N,2) [time, hash] query rows; the hit rows stay on the device."""
        self.tb._sync_device()
        rows = np.ascontiguousarray(np.asarray(hashes, dtype=np.int32).reshape(-1, 2))
        nh = C.c_int64()
        _lib.check(self.lib.afp_table_get_hits(self.h, rows.ctypes.data_as(C.POINTER(C.c_int32)), rows.shape[0],
                                               C.byref(nh)), 'afp_table_get_hits')
        self.nhits = int(nh.value)
        return self.nhits

    def hits(self):
        out = np.zeros((self.nhits, 4), dtype=np.int32)
        _lib.check(self.lib.afp_table_fetch_hits(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))), 'afp_table_fetch_hits')
        return out

    def id_counts(self):
        """(np.unique(hits[:, 0]), np.bincount(hits[:, 0])[ids])  -- audfprint_match.py:128-132."""
        n = C.c_int64()
        _lib.check(self.lib.afp_table_count_ids(self.h, C.byref(n)), 'afp_table_count_ids')
        ids = np.zeros(n.value, dtype=np.int32)
        cnt = np.zeros(n.value, dtype=np.int32)
        I32 = C.POINTER(C.c_int32)
        _lib.check(self.lib.afp_table_fetch_id_counts(self.h, ids.ctypes.data_as(I32), cnt.ctypes.data_as(I32)),
                   'afp_table_fetch_id_counts')
        return ids, cnt

    def skew_hist(self, ids):
        """(mintime, hist[len(ids)][width]) with hist[i][d] = #hits of ids[i] at skew mintime + d  (:281-289)."""
        want = np.ascontiguousarray(ids, dtype=np.int32)
        mt, wd = C.c_int32(), C.c_int32()
        I32 = C.POINTER(C.c_int32)
        _lib.check(self.lib.afp_table_skew_hist(self.h, want.ctypes.data_as(I32), len(want), C.byref(mt), C.byref(wd)),
                   'afp_table_skew_hist')
        hist = np.zeros((len(want), wd.value), dtype=np.int32)
        _lib.check(self.lib.afp_table_fetch_skew_hist(self.h, hist.ctypes.data_as(I32)), 'afp_table_fetch_skew_hist')
        return int(mt.value), hist

    # ---- Matcher._best_count_ids (audfprint_match.py:124-147) ----------------------------------------
    def best_count_ids(self, hashesperid, threshcount, search_depth):
        ids, raw = self.id_counts()
        raw = raw.astype(np.int64)                                   # np.bincount gives int64
        weighted = raw / np.asarray(hashesperid)[ids].astype(float)  # :136
        order = np.argsort(weighted)[::-1]                           # :139
        depth = np.minimum(np.count_nonzero(np.greater(raw, threshcount)), search_depth)   # :142-144
        order = order[:depth]
        return ids[order], raw[order]

    # ---- Matcher._approx_match_counts (audfprint_match.py:241-312), find_time_range off ---------------
    def approx_match_counts(self, ids, rawcounts, window, threshcount, max_alignments_per_id):
        results = np.zeros((len(ids), 7), np.int32)
        if self.nhits == 0:
            return results                                           # :266-268
        mintime, hist = self.skew_hist(ids)
        n = 0
        for urank, (id_, rawcount) in enumerate(zip(ids, rawcounts)):
            row = hist[urank]
            nz = np.nonzero(row)[0]
            bincounts = row[:nz[-1] + 1].astype(np.int64)            # == np.bincount(alltimes[allids == id])
            kept = np.zeros(bincounts.shape)                         # keep_local_maxes (:70-75)
            peaks = np.nonzero(locmax(bincounts))[0]
            kept[peaks] = bincounts[peaks]
            found = 0
            while True:
                mode = int(np.argmax(kept))
                if kept[mode] <= threshcount:                        # :294-297
                    break
                lo, hi = max(0, mode - window), mode + window + 1
                count = np.sum(bincounts[lo:hi])
                results[n, :] = [int(id_), count, mode + mintime, rawcount, urank, 0, 0]
                n += 1
                if n >= results.shape[0]:
                    results = np.vstack([results, np.zeros(results.shape, np.int32)])
                kept[lo:hi] = 0                                      # :308-309
                found += 1
                if found > max_alignments_per_id:                    # :311-312
                    break
        return results[:n, :]


def match_hashes(matcher, tb, hashes, hashesfor=None):
    """``matcher.match_hashes(tb.ht, hashes, hashesfor)`` with the table on the GPU."""
    vc = VoteCounter(tb)
    vc.query(hashes)
    bestids, rawcounts = vc.best_count_ids(tb.ht.hashesperid, matcher.threshcount, matcher.search_depth)
    need_rows = matcher.exact_count or matcher.find_time_range or hashesfor is not None
    hits = vc.hits() if need_rows else None
    if not matcher.exact_count and not matcher.find_time_range:
        results = vc.approx_match_counts(bestids, rawcounts, matcher.window, matcher.threshcount,
                                         matcher.max_alignments_per_id)
    elif not matcher.exact_count:
        results = matcher._approx_match_counts(hits, bestids, rawcounts)
    else:
        results = matcher._exact_match_counts(hits, bestids, rawcounts, hashesfor)
    results = results[(-results[:, 1]).argsort(), ]                  # :336
    if hashesfor is None:
        return results
    return results, matcher._unique_match_hashes(results[hashesfor, 0], hits, results[hashesfor, 2])
