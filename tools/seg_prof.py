#!/usr/bin/env python
"""GPU box: per-kernel times of single-clip extraction (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import afp_oracle as O
from audfprint_amd.batch import Extractor
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
ex = Extractor.get(0)
ex.set_params()
d = O.synth_noise(0, secs)
for rep in range(20):
    r = ex.extract(clips=[d], want_hashes=True, want_peaks=False)
print(ex.seg_stats(), len(r.hashes))
