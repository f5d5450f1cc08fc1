#!/usr/bin/env python
"""Golden fixture for the matcher's vote counting FROM THE LIVE REFERENCE:
hash_table.HashTable filled with the hashes of 14 synthetic tracks, audfprint_match.Matcher.match_hashes run
on excerpts of those tracks (re-analysed from the cut audio, so query hashes are the partly different set a
real query gives), on a mix of two tracks, on an unrelated clip and on an empty query -- for the default
matcher and for non-default window / threshcount / search_depth / max_alignments_per_id.
Run in the build container:  python tests/golden/make_golden_match.py"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import hash_table as RHT  # noqa: E402  (the reference, unchanged)
import audfprint_match as RM  # noqa: E402
from oracle import afp_oracle as O  # noqa: E402

SR = 11025


def main():
    ntracks, secs = 14, 24.0
    audio = [O.synth_tonal(5100 + i, secs) if i % 4 == 0 else O.synth_noise(5100 + i, secs) for i in range(ntracks)]
    track_hashes = [O.extract(a)[1] for a in audio]
    names = ['track%02d.wav' % i for i in range(ntracks)]
    # the same track stored again under another name: two ids with identical votes (a tie for the argsort)
    track_hashes.append(track_hashes[3])
    names.append('track03_copy.wav')
    random.seed(4321)
    ht = RHT.HashTable(hashbits=20, depth=100, maxtime=16384)
    for nm, h in zip(names, track_hashes):
        ht.store(nm, h)
    rng = np.random.RandomState(99)
    queries = []
    for tr, t0, dur, snr in ((3, 5.0, 8.0, 30.0), (8, 11.3, 6.0, 12.0), (0, 2.0, 10.0, 20.0), (11, 0.0, 24.0, 60.0)):
        seg = audio[tr][int(t0 * SR):int((t0 + dur) * SR)].astype(np.float64)
        noise = rng.randn(len(seg)) * np.std(seg) * 10 ** (-snr / 20.0)
        queries.append(O.extract((seg + noise).astype(np.float32), O.Params(shifts=4))[1])
    mix = (0.6 * audio[5][int(4 * SR):int(12 * SR)] + 0.6 * audio[9][int(9 * SR):int(17 * SR)]).astype(np.float32)
    queries.append(O.extract(mix, O.Params(shifts=4))[1])
    queries.append(O.extract(O.synth_noise(777, 8.0), O.Params(shifts=4))[1])       # unrelated
    queries.append(np.zeros((0, 2), np.int32))                                        # empty
    out = dict(rows=np.concatenate(track_hashes).astype(np.int32),
               offsets=np.cumsum([0] + [len(h) for h in track_hashes]).astype(np.int64), names=np.array(names),
               nqueries=np.int32(len(queries)))
    settings = [dict(), dict(window=2, threshcount=3), dict(search_depth=2, max_alignments_per_id=0), dict(threshcount=20)]
    out['nsettings'] = np.int32(len(settings))
    for si, kw in enumerate(settings):
        m = RM.Matcher()
        for k, v in kw.items():
            setattr(m, k, v)
        out['set%d' % si] = np.array([m.window, m.threshcount, m.search_depth, m.max_alignments_per_id], np.int32)
        for qi, q in enumerate(queries):
            hits = ht.get_hits(q)
            ids, raw = m._best_count_ids(hits, ht)
            res = m.match_hashes(ht, q)
            if si == 0:
                out['q%d' % qi] = np.asarray(q, np.int32).reshape(-1, 2)
            out['s%d_q%d_ids' % (si, qi)] = np.asarray(ids, np.int64)
            out['s%d_q%d_raw' % (si, qi)] = np.asarray(raw, np.int64)
            out['s%d_q%d_res' % (si, qi)] = np.asarray(res, np.int32).reshape(-1, 7)
            # the remaining modes of the matcher -- exact counts (:195-239) and time ranges (:173-193)
            for ec, tr in ((1, 0), (1, 1), (0, 1)):
                m2 = RM.Matcher()
                for k, v in kw.items():
                    setattr(m2, k, v)
                m2.exact_count, m2.find_time_range = bool(ec), bool(tr)
                if len(q) == 0 or len(hits) == 0:
                    r2 = np.zeros((0, 7), np.int32)          # (the reference's argsort of an empty hit list raises nothing, its modes loop finds nothing)
                    try:
                        r2 = m2.match_hashes(ht, q)
                    except Exception:
                        pass
                else:
                    r2 = m2.match_hashes(ht, q)
                out['s%d_q%d_res_e%d_t%d' % (si, qi, ec, tr)] = np.asarray(r2, np.int32).reshape(-1, 7)
            print('setting', si, 'query', qi, 'hashes', len(q), 'hits', len(hits), 'cands', len(ids), 'results',
                  np.asarray(res).reshape(-1, 7)[:2].tolist())
    np.savez_compressed(os.path.join(HERE, 'match_votes.npz'), **out)


if __name__ == '__main__':
    main()
