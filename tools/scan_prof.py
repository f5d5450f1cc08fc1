#!/usr/bin/env python
"""GPU box: k_scan phase breakdown from the debug cycle stamps (tap 5)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import afp_oracle as O
from audfprint_amd.batch import Extractor
ex = Extractor.get(0)
ex.set_params()
for nclips, secs in ((1, 300.0), (1024, 30.0)):
    pool = [O.synth_noise(i, secs) for i in range(min(nclips, 16))]
    clips = [pool[i % len(pool)] for i in range(nclips)]
    pcm, off = Extractor.pack(clips)
    for rep in range(2):
        r = ex.extract(pcm=pcm, offsets=off, want_hashes=True, debug=not os.environ.get('AFP_SCAN_PROF'))
    p = ex.debug(5, np.uint64, (32,)).astype(np.int64)
    d = np.diff(p[:, :6], axis=1)
    T = p[:, 6]
    names = ['wait B0', 'preroll+init', 'forward', 'bwd init', 'backward']
    print('nclips=%d secs=%g  T=%d' % (nclips, secs, T[0]))
    for i, nm in enumerate(names):
        print('   %-14s mean %10.0f cycles   per-frame %8.1f' % (nm, d[:, i].mean(), (d[:, i] / T).mean()))
    fw = (p[:, 7] >> 32) & 0xffffffff; bw = p[:, 7] & 0xffffffff
    print('   scanner barrier wait: fwd %.1f cycles/frame, bwd %.1f cycles/frame' % ((fw / T).mean(), (bw / T).mean()))
    tot = p[:, 12:15].sum()
    print('   fwd frame classes: zero %.1f%% (%.0f cyc), one %.1f%% (%.0f cyc), multi %.1f%% (%.0f cyc); LDS frame read %.0f cyc/frame' % (
        100.0 * p[:, 12].sum() / tot, p[:, 9].sum() / max(1, p[:, 12].sum()), 100.0 * p[:, 13].sum() / tot, p[:, 10].sum() / max(1, p[:, 13].sum()),
        100.0 * p[:, 14].sum() / tot, p[:, 11].sum() / max(1, p[:, 14].sum()), p[:, 8].sum() / tot))
    print('   bwd frame classes: empty %.1f%% (%.0f cyc), with records %.1f%% (%.0f cyc; %.2f records/frame, %.0f%% kept)' % (
        100.0 * p[:, 18].sum() / max(1, (p[:, 18] + p[:, 19]).sum()), p[:, 16].sum() / max(1, p[:, 18].sum()),
        100.0 * p[:, 19].sum() / max(1, (p[:, 18] + p[:, 19]).sum()), p[:, 17].sum() / max(1, p[:, 19].sum()),
        p[:, 20].sum() / max(1, p[:, 19].sum()), 100.0 * p[:, 21].sum() / max(1, p[:, 20].sum())))
    print('   span of starts: %d cycles; total mean %d' % (p[:, 0].max() - p[:, 0].min(), (p[:, 5] - p[:, 0]).mean()))
