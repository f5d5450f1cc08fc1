#!/usr/bin/env python
"""GPU box: where the time of ONE file through the drop-in goes (the call pattern of audfprint.py:164-165, 177-182).
Prints, for 10 / 30 / 60 / 300 s clips: ms per Extractor.extract call with and without the peak list, the host phases of
one call (pack, the enqueueing C call, the wait for the counts, the row fetch), the kernels' own time (HIP events), and the
segment-parallel scan forced onto short clips with several (segment length, warm-up) pairs -- each checked against the
default path's rows.   Usage: python tools/analyzer_breakdown.py [secs ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import afp_oracle as O                      # noqa: E402  (synthetic clips only)
from audfprint_amd import _lib                          # noqa: E402
from audfprint_amd.batch import Extractor               # noqa: E402


def timed(fn, n):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    return (time.perf_counter() - t0) / n * 1e3, r


def main():
    secs_list = [float(a) for a in sys.argv[1:]] or [10.0, 30.0, 60.0, 300.0]
    ex = Extractor.get(0)
    ex.set_params()
    lib = ex.lib
    I32, I64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    for secs in secs_list:
        d = np.ascontiguousarray(O.synth_noise(77, secs), dtype=np.float32)
        n = 100 if secs <= 60 else 20
        ex.set_pipeline()
        ms_hp, r = timed(lambda: ex.extract(clips=[d], want_hashes=True, want_peaks=True), n)
        ms_h, r0 = timed(lambda: ex.extract(clips=[d], want_hashes=True, want_peaks=False), n)
        print('%5.0f s clip (%d frames): extract hashes+peaks %.3f ms, hashes only %.3f ms; %d hashes; seg %s'
              % (secs, 1 + len(d) // 256, ms_hp, ms_h, len(r0.hashes), ex.seg_stats()))
        # host phases of one hashes+peaks call
        ph = np.zeros(6)
        for rep in range(n + 3):
            t = [time.perf_counter()]
            pcm, offsets = ex.pack([d], np.float32)
            t.append(time.perf_counter())
            _lib.check(lib.afp_extract_host(ex.h, pcm.ctypes.data_as(C.POINTER(C.c_float)), offsets.ctypes.data_as(I64), 1, 3))
            t.append(time.perf_counter())
            th, tp, nu = ex.counts()
            t.append(time.perf_counter())
            hs = np.empty((th, 2), np.int32); ho = np.zeros(2, np.int64)
            pk = np.empty((tp, 2), np.int32); po = np.zeros(nu + 1, np.int64); fl = np.zeros(nu, np.int32)
            t.append(time.perf_counter())
            _lib.check(lib.afp_fetch_all(ex.h, hs.ctypes.data_as(I32), ho.ctypes.data_as(I64), pk.ctypes.data_as(I32),
                                         po.ctypes.data_as(I64), fl.ctypes.data_as(I32)))
            t.append(time.perf_counter())
            if rep >= 3:
                ph[:5] += np.diff(t)
        ph = ph / n * 1e3
        print('        phases: pack %.3f | enqueue (H2D + launches) %.3f | wait for counts %.3f | alloc %.3f | fetch rows %.3f ms'
              % tuple(ph[:5]))
        ex.set_timing(True)
        ex.reset_timings()
        for _ in range(10):
            ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        tm = ex.timings()
        ex.set_timing(False)
        print('        kernels (HIP events, ms): ' + ', '.join('%s %.4f' % (k, v[0] / max(v[1], 1)) for k, v in tm.items() if v[1]))
        # the shader clock the chip holds while nothing but this one-file-at-a-time loop runs on it
        try:
            ex.clock_probe_start(20)
            tend = time.perf_counter() + 0.03
            while time.perf_counter() < tend:
                ex.extract(clips=[d], want_hashes=True, want_peaks=False)
            print('        shader clock during the loop: %.0f MHz' % ex.clock_probe_stop())
        except Exception as e:              # noqa: BLE001
            print('        clock probe failed: %r' % (e,))
        if os.environ.get('AFP_BREAKDOWN_NO_SEG_SWEEP'):
            continue
        # the segment-parallel scan on this clip, forced, with shorter segments / warm-ups
        for L, W in ((0, 0), (104, 205), (64, 128), (48, 96), (80, 160)):
            ex.set_pipeline(seg=1, seg_len=L, seg_warm=W)
            ms, rs = timed(lambda: ex.extract(clips=[d], want_hashes=True, want_peaks=False), n)
            st = ex.seg_stats()
            print('        seg forced L=%3d W=%3d: %.3f ms  same rows %s  %s'
                  % (L, W, ms, bool(np.array_equal(rs.hashes, r0.hashes)), st))
        ex.set_pipeline(seg=0)
        ms, rs = timed(lambda: ex.extract(clips=[d], want_hashes=True, want_peaks=False), n)
        print('        seg off: %.3f ms  same rows %s' % (ms, bool(np.array_equal(rs.hashes, r0.hashes))))
        ex.set_pipeline()


if __name__ == '__main__':
    main()
