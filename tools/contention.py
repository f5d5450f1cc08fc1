#!/usr/bin/env python
"""What stretches k_stft when it runs beside something else?  (GPU box)

Times the spectral stage (k_stft + stats + floor correction) of a C3 batch (a) alone, (b) beside a pure HBM
stream (device-to-device copies on another stream: no VALU to speak of), (c) beside a VALU-only kernel (torch
elementwise chain on a tiny, cache-resident tensor: no HBM traffic), (d) beside the scan stage of another
batch (the real pipeline).  Wall clock over N repetitions, HIP events on the spectral stream."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audfprint_amd.batch import Extractor  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    nclips, nsamp = 1024, 30 * 11025
    pool = bench.synth_pool(64, nsamp, 7)
    d_pcm = torch.from_numpy(pool).to(dev).repeat(16, 1).contiguous().view(-1)
    offsets = np.arange(nclips + 1, dtype=np.int64) * nsamp
    ex = Extractor(0)
    ex.set_params()
    ex2 = Extractor(0)
    ex2.set_params()
    s_main = torch.cuda.Stream(device=dev)
    s_side = torch.cuda.Stream(device=dev)
    ex.set_stream(s_main.cuda_stream)
    ex2.set_stream(s_side.cuda_stream)
    for e in (ex, ex2):
        e.extract_device(d_pcm.data_ptr(), offsets)
        e.counts()
    big_a = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB
    big_b = torch.empty_like(big_a)
    small = torch.zeros(16, dtype=torch.float64, device=dev)
    import ctypes as C
    import subprocess
    so = '/tmp/libhog.so'
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(ROOT, 'tools', 'hog.hip'), '-o', so])
    hog = C.CDLL(so)
    hog.hog_valu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    hog.hog_hbm.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def run(side):
        ex.set_timing(True)
        ex.reset_timings()
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            if side == 'hbm':
                hog.hog_hbm(big_a.data_ptr(), big_b.data_ptr(), big_a.numel() * 4, 2048, s_side.cuda_stream)   # 2 GiB moved: ~0.4 ms
                hog.hog_hbm(big_a.data_ptr(), big_b.data_ptr(), big_a.numel() * 4, 2048, s_side.cuda_stream)
                hog.hog_hbm(big_a.data_ptr(), big_b.data_ptr(), big_a.numel() * 4, 2048, s_side.cuda_stream)
            elif side == 'valu':
                hog.hog_valu(small.data_ptr(), 512, 60000, s_side.cuda_stream)     # 2 waves/SIMD of FP64 FMA for ~1 ms
            elif side == 'scan':
                ex2.extract_device(d_pcm.data_ptr(), offsets)
            ex.extract_device(d_pcm.data_ptr(), offsets)
            ex.counts()
            torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        tm = ex.timings()
        ex.set_timing(False)
        return dict(wall_ms=round(wall, 3), **{k: round(v[0] / max(1, v[1]), 4) for k, v in tm.items() if v[1]})

    out = {}
    for side in ('alone', 'hbm', 'valu', 'scan', 'alone'):
        out[side + ('' if side not in out else '_again')] = run(None if side == 'alone' else side)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
