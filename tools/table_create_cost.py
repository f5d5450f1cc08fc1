#!/usr/bin/env python
"""GPU box: where do the milliseconds of TableBuilder's creation go (bench.py c4_job counts it inside the job since round 6)?"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audfprint_amd                                   # noqa: E402
audfprint_amd.configure_runtime()
from audfprint_amd import _lib                         # noqa: E402
from audfprint_amd.batch import Extractor              # noqa: E402
from audfprint_amd.table import TableBuilder           # noqa: E402


class HT(object):
    def __init__(self):
        self.hashbits, self.depth, self.maxtimebits = 20, 100, 14
        self.table = np.zeros((1 << 20, 100), np.uint32)
        self.counts = np.zeros(1 << 20, np.int32)
        self.names, self.hashesperid = [], np.zeros(0, np.uint32)


ex = Extractor.get(0)
lib = _lib.load()
for rep in range(4):
    ht = HT()
    t0 = time.perf_counter()
    _lib.check(lib.afp_table_create(ex.h, 20, 100, 14))
    t1 = time.perf_counter()
    nz = int(np.count_nonzero(ht.counts))
    t2 = time.perf_counter()
    n = lib.afp_host_prefault(C.c_void_p(ht.table.ctypes.data), ht.table.nbytes) if rep % 2 == 0 else 0
    t3 = time.perf_counter()
    import torch
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    tb = TableBuilder(HT(), ex, prefault=bool(rep % 2 == 0))
    t5 = time.perf_counter()
    print('rep %d: afp_table_create %.3f ms | count_nonzero(counts) %.3f | prefault start (%d threads) %.3f | device sync after %.3f | '
          'whole TableBuilder() %.3f ms' % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
    time.sleep(0.05)
