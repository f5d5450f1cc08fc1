#!/usr/bin/env python
"""GPU box: segment-parallel scan vs the sequential kernel vs the oracle on long single clips."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import afp_oracle as O
from audfprint_amd.batch import Extractor
ex = Extractor.get(0)
for name, d, kw in (('noise300', O.synth_noise(0, 300.0), {}), ('tonal120', O.synth_tonal(5, 120.0), {}),
                    ('noise60 d70 s4', O.synth_noise(2, 60.0), dict(density=70.0, maxpairsperpeak=10, shifts=4)),
                    ('noise10', O.synth_noise(1, 10.0), {})):
    ex.set_params(**kw)
    prm = O.Params(**kw)
    pls, hs = O.extract(d, prm)
    for rep in range(3):
        t0 = time.perf_counter()
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        dt = time.perf_counter() - t0
    ok = np.array_equal(r.clip_hashes(0), hs) and all(np.array_equal(r.unit_peaks(0, s), pls[s]) for s in range(prm.shifts))
    print(name, 'seg', ex.seg_stats(), 'bit_exact', ok, 'hashes', len(hs), 'wall ms', round(dt * 1e3, 3))
