#!/usr/bin/env python
"""GPU box, AFP_HPF_PROF=1: cycle stamps of k_hpf's filter wavefront (workgroup 0 of unit 0) per phase of 32 frames --
how long it works, how long a phase lasts (the difference is the wait at the barrier for the loader)."""
import os
import sys
import numpy as np
os.environ['AFP_HPF_PROF'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import afp_oracle as O
from audfprint_amd.batch import Extractor
ex = Extractor.get(0)
ex.set_params()
d = O.synth_noise(0, 300.0)
for rep in range(3):
    r = ex.extract(clips=[d], want_hashes=True, want_peaks=False)
st = ex.debug(6, np.uint64).reshape(-1, 4).astype(np.int64)
n = int(np.max(np.nonzero(st[:, 0])[0])) + 1
st = st[:n]
f_work = st[:, 1] - st[:, 0]
f_next = np.diff(st[:, 0])
print('phases stamped', n, ' filter wavefront: work %.0f cycles / phase of 32 frames (median; fast phases %.0f, phases with a listed frame %.0f), '
      'phase period %.0f' % (np.median(f_work), np.percentile(f_work, 10), np.percentile(f_work, 90), np.median(f_next)))
