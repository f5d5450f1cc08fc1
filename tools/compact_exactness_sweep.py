#!/usr/bin/env python
"""GPU box: how often does the COMPACT spectral stage (the path every large batch takes; its filtered values differ from the
dense path's by a few ulps because the per-unit mean is subtracted after the onset filter) give other integers than the DENSE
path, which follows the reference's order of operations?  DESIGN.md §2 calls compact-path exactness "a property established by
volume"; this is the volume: millions of distinct clips generated on the device (the SURVEY §8c recipe -- Gaussian noise,
sigma 0.1, clipped, quantised to the int16 grid -- with torch's generator; every fourth batch mixes in gated tones and
per-clip levels from -42 dB up), each batch run three times:
    compact, guard off   -- the production kernels (what bench.py times)
    dense                -- the reference's arithmetic order; rows compared with the compact rows, clip by clip
    compact, guard 1e-11 -- counts the units in which a decisive comparison was closer than 1e-11 (AFP_UNIT_NEARTIE)

    python tools/compact_exactness_sweep.py [--batches 1000] [--clips 1024] [--secs 30] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audfprint_amd                                    # noqa: E402
audfprint_amd.configure_runtime()
import torch                                            # noqa: E402
from audfprint_amd import _lib                          # noqa: E402
from audfprint_amd.batch import Extractor               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, default=1000)
    ap.add_argument('--clips', type=int, default=1024)
    ap.add_argument('--secs', type=float, default=30.0)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--density', type=float, default=20.0)
    ap.add_argument('--shifts', type=int, default=1)
    ap.add_argument('--max-seconds', type=float, default=1500.0)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(a.seed)
    ex = Extractor.get(0)
    ex.set_params(density=a.density, maxpairsperpeak=3 if a.density < 50 else 10, shifts=a.shifts)
    n = int(round(a.secs * 11025))
    off = np.arange(a.clips + 1, dtype=np.int64) * n
    t = torch.arange(n, device=dev, dtype=torch.float32) / 11025.0
    units = clips = bad_batches = bad_clips = near = redone = rows = 0
    t0 = time.time()
    print('build %s | %d batches x %d clips x %.0f s, density %.0f, shifts %d, seed %d' %
          (_lib.load().afp_build_id().decode(), a.batches, a.clips, a.secs, a.density, a.shifts, a.seed), flush=True)
    for b in range(a.batches):
        if time.time() - t0 > a.max_seconds:
            break
        x = torch.randn((a.clips, n), generator=gen, device=dev) * 0.1
        if b % 4 == 3:
            # gated tones + per-clip levels: plateaus, digital-silence-like stretches, quiet clips
            f0 = torch.rand((a.clips, 1), generator=gen, device=dev) * 3000.0 + 100.0
            gate = (torch.sin(2 * np.pi * 2.0 * t)[None, :] > 0).float()
            x = 0.02 * x + 0.3 * torch.sin(2 * np.pi * f0 * t[None, :]) * gate
            lvl = 10.0 ** (-(torch.rand((a.clips, 1), generator=gen, device=dev) * 42.0) / 20.0)
            x = x * lvl
        pcm = (torch.round(torch.clamp(x, -1, 1) * 32767.0) / 32768.0).contiguous().view(-1)
        del x
        torch.cuda.synchronize()          # (the library runs on its own streams: the clips must be in HBM before it reads them)
        res = {}
        for name, kw, eps in (('compact', dict(compact=1, seg=0), 0.0), ('dense', dict(compact=0, seg=0), 0.0), ('guarded', dict(compact=1, seg=0), 1e-11)):
            ex.set_pipeline(**kw)
            ex.set_neartie_eps(eps)
            ex.extract_device(pcm.data_ptr(), off, want_hashes=True, want_peaks=False)
            r = ex.fetch(a.clips, True, False)
            ps = ex.path_stats()
            if name == 'guarded':
                near += int(np.count_nonzero(r.unit_flags & _lib.UNIT_NEARTIE)) if not ps['near_tie_redone'] else ps['near_tie_units']
                redone += 1 if ps['near_tie_redone'] else 0
            elif name == 'compact' and (not ps['compact'] or ps['redone_dense']):
                print('batch %d: the compact pass did not run compact: %s' % (b, ps), flush=True)
            res[name] = r
        ex.set_neartie_eps(0.0)
        c, d = res['compact'], res['dense']
        if not (np.array_equal(c.hash_offsets, d.hash_offsets) and np.array_equal(c.hashes, d.hashes)):
            bad_batches += 1
            for i in range(a.clips):
                if not np.array_equal(c.clip_hashes(i), d.clip_hashes(i)):
                    bad_clips += 1
            print('MISMATCH compact vs dense in batch %d' % b, flush=True)
        units += a.clips * a.shifts
        clips += a.clips
        rows += len(d.hashes)
        if (b + 1) % 100 == 0:
            print('  %d batches, %d clips, %.0f s: mismatching clips %d, near-tie units (1e-11) %d' % (b + 1, clips, time.time() - t0, bad_clips, near), flush=True)
    ex.set_pipeline()
    print('RESULT %d clips (%d units, %.3g audio-seconds, %d rows): clips whose compact rows differ from the dense rows: %d (in %d batches); '
          'units with a decisive comparison closer than 1e-11: %d (compact batches re-run densely for that: %d); %.0f s' %
          (clips, units, clips * a.secs, rows, bad_clips, bad_batches, near, redone, time.time() - t0))
    return 1 if bad_clips else 0


if __name__ == '__main__':
    sys.exit(main())
