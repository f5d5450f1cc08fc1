"""Query-side vote counting on the GPU ("next" row f4).

``match_hashes(matcher, tb, hashes)`` gives what ``Matcher.match_hashes(ht, hashes)``
(audfprint_match.py:314-352) gives, with the table walk and the histogram work done on the device-resident
table of a ``TableBuilder``:

* ``HashTable.get_hits`` (hash_table.py:150-176)                       -> ``afp_table_get_hits``
* ``np.unique`` / ``np.bincount`` of ``_best_count_ids`` (:128-132)    -> ``afp_table_count_ids``
* the per-id ``np.bincount`` loop of ``_approx_match_counts`` (:289)   -> ``afp_table_skew_hist``

* the row selections of ``_exact_match_counts`` / ``_unique_match_hashes`` / ``_calculate_time_ranges``
  (:149-239: ``allids == id`` and ``abs(alltimes - mode) <= window``), all candidate alignments in one pass
                                                                       -> ``afp_table_select_hits``

What stays on the host is a few numpy calls over arrays the size of the candidate list (the weighting and
``np.argsort`` of :133-147 -- numpy's own sort decides ties, so it has to be numpy's -- the mode picking of
:78-90 / :291-311, and ``np.unique`` / quantile picks over the few selected rows of an alignment).  ``matcher`` is the
caller's ``audfprint_match.Matcher`` (or anything with its attributes ``window, threshcount, search_depth,
max_alignments_per_id, exact_count, find_time_range, time_quantile``); only ``hashesfor`` (the matching hashes of one
result, for display) downloads the hit rows and runs the matcher's own ``_unique_match_hashes`` on them.
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

    def max_orig_time(self):
        """np.amax(hits[:, 3]) (after id_counts): sizes the packed keys of _unique_match_hashes (:157)."""
        v = C.c_int32()
        _lib.check(self.lib.afp_table_hits_max_time(self.h, C.byref(v)), 'afp_table_hits_max_time')
        return int(v.value)

    def select(self, ids, lo, hi):
        """For every query q the (orig_time, hash) rows of the hits with id == ids[q] and lo[q] <= skew <= hi[q], as a list of
        (n_q, 2) int32 arrays -- the selections of :159-163 / :181-185 for all candidate alignments in one pass over the hits
        in HBM.  Row order inside a query is unspecified (the callers take np.unique / np.sort)."""
        I32 = C.POINTER(C.c_int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        nq = len(ids)
        tot = C.c_int64()
        _lib.check(self.lib.afp_table_select_hits(self.h, ids.ctypes.data_as(I32), lo.ctypes.data_as(I32), hi.ctypes.data_as(I32), nq,
                                                  C.byref(tot)), 'afp_table_select_hits')
        rows = np.zeros((tot.value, 2), dtype=np.int32)
        off = np.zeros(nq + 1, dtype=np.int64)
        _lib.check(self.lib.afp_table_fetch_selected(self.h, rows.ctypes.data_as(I32), off.ctypes.data_as(C.POINTER(C.c_int64))),
                   'afp_table_fetch_selected')
        return [rows[off[q]:off[q + 1]] for q in range(nq)]

    # ---- Matcher._unique_match_hashes (audfprint_match.py:149-171) from rows selected on the device -------
    def unique_match_hashes(self, id_, mode, window):
        """The (time, hash) rows of the query hashes that support alignment (id_, mode): hits with that id and
        |skew - mode| <= window (:159-163), uniquified through the packed key orig_time + (hash << timebits) (:164-171) -- what
        `match_hashes(..., hashesfor=k)` returns for result k (`audfprint.py match --illustrate`).  The selection runs on the
        device (afp_table_select_hits); only the selected rows come back."""
        rows = self.select([int(id_)], [int(mode) - int(window)], [int(mode) + int(window)])[0]
        timebits = max(1, int(np.ceil(np.log(max(1, self.max_orig_time())) / np.log(2))))      # :157 (encpowerof2 :45-47)
        key = np.unique(rows[:, 0].astype(np.int64) + (rows[:, 1].astype(np.int64) << timebits))
        return np.c_[key & ((1 << timebits) - 1), key >> timebits]

    # ---- Matcher._calculate_time_ranges (audfprint_match.py:173-193) over selected rows ---------------
    @staticmethod
    def _time_range(sel_rows, time_quantile):
        match_times = np.sort(sel_rows[:, 0])                        # the reference reads them off hits sorted by orig_time (:208, :181-185)
        return (match_times[int(len(match_times) * time_quantile)],
                match_times[int(len(match_times) * (1.0 - time_quantile)) - 1])

    # ---- Matcher._exact_match_counts (audfprint_match.py:195-239) --------------------------------------
    def exact_match_counts(self, ids, rawcounts, window, threshcount, find_time_range=False, time_quantile=0.02):
        results = np.zeros((0, 7), np.int32)
        if self.nhits == 0 or len(ids) == 0:
            return results
        mintime, hist = self.skew_hist(ids)
        # find_modes (:78-90) per id on ITS skew histogram: np.bincount(data - min(data)) is the id's row from its first non-zero bin
        cand = []                                                    # (urank, id, mode)
        for urank, id_ in enumerate(ids):
            row = hist[urank]
            nz = np.nonzero(row)[0]
            full = row[nz[0]:nz[-1] + 1].astype(np.int64)
            datamin = mintime + int(nz[0])
            modes = np.nonzero(np.logical_and(locmax(full), np.greater_equal(full, threshcount)))[0] + datamin
            cand.extend((urank, int(id_), int(m)) for m in modes)
        if not cand:
            return results
        sel = self.select([c[1] for c in cand], [c[2] - window for c in cand], [c[2] + window for c in cand])
        timebits = max(1, int(np.ceil(np.log(max(1, self.max_orig_time())) / np.log(2))))      # :157 (encpowerof2 :45-47)
        out = []
        min_time = max_time = 0
        for (urank, id_, mode), rows in zip(cand, sel):
            # len(np.unique(allotimes[matchix] + (allhashes[matchix] << timebits)))   :166-167
            filtcount = len(np.unique(rows[:, 0].astype(np.int64) + (rows[:, 1].astype(np.int64) << timebits)))
            if filtcount >= threshcount:                             # :224
                if find_time_range:
                    min_time, max_time = self._time_range(rows, time_quantile)
                out.append([id_, filtcount, mode, int(rawcounts[urank]), urank, min_time, max_time])
        return np.array(out, dtype=np.int32).reshape(-1, 7)

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
    def approx_match_counts(self, ids, rawcounts, window, threshcount, max_alignments_per_id, find_time_range=False,
                            time_quantile=0.02):
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
        results = results[:n, :]
        if find_time_range and n:                                    # :300-302, all rows in one selection pass
            sel = self.select(results[:, 0], results[:, 2] - window, results[:, 2] + window)
            for r, rows in enumerate(sel):
                results[r, 5:7] = self._time_range(rows, time_quantile)
        return results


def match_hashes(matcher, tb, hashes, hashesfor=None):
    """``matcher.match_hashes(tb.ht, hashes, hashesfor)`` with the table on the GPU."""
    vc = VoteCounter(tb)
    vc.query(hashes)
    bestids, rawcounts = vc.best_count_ids(tb.ht.hashesperid, matcher.threshcount, matcher.search_depth)
    if not matcher.exact_count:
        results = vc.approx_match_counts(bestids, rawcounts, matcher.window, matcher.threshcount, matcher.max_alignments_per_id,
                                         bool(matcher.find_time_range), getattr(matcher, 'time_quantile', 0.02))
    else:
        results = vc.exact_match_counts(bestids, rawcounts, matcher.window, matcher.threshcount, bool(matcher.find_time_range),
                                        getattr(matcher, 'time_quantile', 0.02))
    results = results[(-results[:, 1]).argsort(), ]                  # :336
    if hashesfor is None:
        return results
    return results, vc.unique_match_hashes(results[hashesfor, 0], results[hashesfor, 2], matcher.window)     # :346-352
