#!/usr/bin/env python
"""Soak of the host machinery around the kernels: the 12 500-clip `new` job's shape -- pinned s16 PCM ->
upload stream -> staged contexts -> device table store with overflow replay -> table download (sparse and dense) -- the C3
pipeline on resident PCM and the one-file path, interleaved, for thousands of iterations, every iteration's results
compared with the first iteration's.  A GPU memory fault kills the process: the log is flushed as it goes, and the
wrapper sets AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1.

  python tools/soak.py --iters 2000                      # as shipped: torch in the process (its bundled HIP runtime), 12 hardware queues
  python tools/soak.py --iters 2000 --no-torch           # the system HIP runtime, no torch: pinned memory from afp_pinned_alloc,
                                                         # streams from hipStreamCreateWithPriority through ctypes
  AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 python tools/soak.py ...     GPU_MAX_HW_QUEUES=4 python tools/soak.py ...
Prints one JSON line at the end (also appended to --log)."""
import argparse
import ctypes as C
import hashlib
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=2000)
ap.add_argument('--no-torch', action='store_true')
ap.add_argument('--clips', type=int, default=5000, help='10 s clips of one job iteration')
ap.add_argument('--batch', type=int, default=1250)
ap.add_argument('--ctx', type=int, default=3)
ap.add_argument('--c3-clips', type=int, default=1024)
ap.add_argument('--c3-steps', type=int, default=3)
ap.add_argument('--reset-every', type=int, default=25, help='iterations between fresh tables (whose content is compared with the first fresh table\'s)')
ap.add_argument('--dense-every', type=int, default=7, help='every n-th download is the whole table (the r04 path), the others move the filled prefixes')
ap.add_argument('--log', default='')
ap.add_argument('--tag', default='')
args = ap.parse_args()

if args.no_torch:
    os.environ['AFP_HIP_RUNTIME'] = 'system'
import numpy as np                                                       # noqa: E402
import audfprint_amd                                                     # noqa: E402
audfprint_amd.configure_runtime()
from audfprint_amd import _lib                                           # noqa: E402
from audfprint_amd.batch import Extractor, pinned_empty                  # noqa: E402
from audfprint_amd.table import TableBuilder                             # noqa: E402

torch = None
if not args.no_torch:
    import torch                                                         # noqa: E402
    assert torch.cuda.is_available()
SR = 11025


def logline(msg):
    line = '[%s %8.1fs] %s' % (args.tag or ('notorch' if args.no_torch else 'torch'), time.perf_counter() - T0, msg)
    print(line, flush=True)
    if args.log:
        with open(args.log, 'a') as f:
            f.write(line + '\n')


class Hip(object):
    """The four runtime calls the torch-free variant needs, through ctypes on the runtime libafp_hip.so bound."""

    def __init__(self):
        self.l = C.CDLL('libamdhip64.so.7')
        self.l.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.l.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.l.hipStreamCreateWithPriority.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_int]
        self.l.hipDeviceGetStreamPriorityRange.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]

    def chk(self, e, what):
        if e != 0:
            raise RuntimeError('%s: hip error %d' % (what, e))

    def malloc(self, n):
        p = C.c_void_p()
        self.chk(self.l.hipMalloc(C.byref(p), n), 'hipMalloc')
        return p.value

    def h2d(self, dptr, arr):
        self.chk(self.l.hipMemcpy(dptr, arr.ctypes.data, arr.nbytes, 1), 'hipMemcpy')

    def stream(self, high):
        lo, hi = C.c_int(), C.c_int()
        self.chk(self.l.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), 'range')
        s = C.c_void_p()
        self.chk(self.l.hipStreamCreateWithPriority(C.byref(s), 1, hi.value if high else 0), 'hipStreamCreateWithPriority')
        return s.value

    def sync(self):
        self.chk(self.l.hipDeviceSynchronize(), 'hipDeviceSynchronize')


class Table(object):
    """The fields of hash_table.HashTable that TableBuilder touches (hash_table.py:61-83)."""

    def __init__(self, hashbits=20, depth=100, maxtime=16384):
        self.hashbits, self.depth, self.maxtimebits = hashbits, depth, 14
        self.table = np.zeros((1 << hashbits, depth), dtype=np.uint32)
        self.counts = np.zeros(1 << hashbits, dtype=np.int32)
        self.names, self.hashesperid, self.dirty = [], np.zeros(0, np.uint32), True


T0 = time.perf_counter()
ex0 = Extractor.get(0)
info = audfprint_amd.runtime_info()
logline('runtime %s' % json.dumps(info))
logline('build %s, host threads %d, env %s' % (_lib.load().afp_build_id().decode(), _lib.load().afp_host_threads(),
                                               {k: v for k, v in os.environ.items() if k.startswith(('AMD_', 'GPU_', 'AFP_', 'HIP_', 'HSA_'))}))
hip = Hip() if args.no_torch else None
# ---- contexts and their stage streams (bench.py Runner.contexts: spectral | scan | pair, the last two high priority) ----
nctx = max(4, args.ctx)
exs = [ex0] + [Extractor(0) for _ in range(nctx - 1)]
if args.no_torch:
    stages = [hip.stream(False), hip.stream(True), hip.stream(True)]
else:
    _ts = [torch.cuda.Stream(device='cuda:0'), torch.cuda.Stream(device='cuda:0', priority=-1), torch.cuda.Stream(device='cuda:0', priority=-1)]
    stages = [s.cuda_stream for s in _ts]
for e in exs:
    e.set_params(density=20.0, maxpairsperpeak=3, shifts=1)
    e.set_stage_streams(*stages)

# ---- inputs: s16 noise, 256 distinct 10 s clips tiled over the job (pinned), 64 distinct 30 s clips tiled over C3 (HBM) ----
rng = np.random.RandomState(12345)
ns10, ns30 = 10 * SR, 30 * SR
pool10 = np.round(np.clip(rng.randn(256, ns10) * 0.1, -1, 1) * 32767).astype(np.int16)
if args.no_torch:
    pin = pinned_empty((args.clips, ns10), np.int16)
else:
    _pin_t = torch.empty((args.clips, ns10), dtype=torch.int16).pin_memory()
    pin = _pin_t.numpy()
for lo in range(0, args.clips, 256):
    hi = min(args.clips, lo + 256)
    pin[lo:hi] = pool10[:hi - lo]
flat = pin.reshape(-1)
pool30 = (np.round(np.clip(rng.randn(64, ns30) * 0.1, -1, 1) * 32767) / 32768.0).astype(np.float32)
c3 = np.concatenate([pool30] * (args.c3_clips // 64), axis=0).reshape(-1)
if args.no_torch:
    d_c3 = hip.malloc(c3.nbytes)
    hip.h2d(d_c3, c3)
    c3_ptr = d_c3
else:
    _d_c3 = torch.from_numpy(c3).to('cuda:0')
    c3_ptr = _d_c3.data_ptr()
off30 = np.arange(args.c3_clips + 1, dtype=np.int64) * ns30
one10 = (pool10[3].astype(np.float32) / np.float32(32768))
one300 = np.concatenate([pool30[i] for i in range(10)])
names_all = ['clip%07d' % i for i in range(args.clips)]
nb = (args.clips + args.batch - 1) // args.batch


def sync_all():
    if args.no_torch:
        hip.sync()
    else:
        torch.cuda.synchronize()


def job(tb, it):
    """One `new` job over the pinned clips: pipelined submits, stores in clip order, then the download."""
    pend, nh, per_batch = [], 0, []
    ctx = exs[:args.ctx]

    def retire():
        e, lo, hi = pend.pop(0)
        off = e.fetch_offsets(hi - lo)
        tb.store_batch(['i%d_%s' % (it, n) for n in names_all[lo:hi]], offsets=off, src=e)
        per_batch.append(int(off[-1]))
        return int(off[-1])
    for b in range(nb):
        lo, hi = b * args.batch, min(args.clips, (b + 1) * args.batch)
        e = ctx[b % len(ctx)]
        if len(pend) == len(ctx):
            nh += retire()
        e.submit(flat[lo * ns10:hi * ns10], np.arange(hi - lo + 1, dtype=np.int64) * ns10)
        pend.append((e, lo, hi))
    while pend:
        nh += retire()
    tb._in_step = tb._in_step_default and (it % args.dense_every != args.dense_every - 1)
    tb.finalize()
    return nh, per_batch


def c3_steps(n):
    fl, tot = [], []
    for i in range(n):
        e = exs[i % len(exs)]
        if len(fl) == len(exs):
            tot.append(fl.pop(0).counts()[0])
        e.extract_device(c3_ptr, off30, want_hashes=True, want_peaks=False)
        fl.append(e)
    for e in fl:
        tot.append(e.counts()[0])
    return tot


def one_file(d):
    r = exs[-1].extract(clips=[d], want_hashes=True, want_peaks=True)
    return hashlib.sha256(r.hashes.tobytes() + r.peaks.tobytes()).hexdigest()[:16]


ht = Table()
tb = None
want = {}
stored_since_reset = 0
mism = []
t_job = t_c3 = t_one = 0.0
for it in range(args.iters):
    fresh = it % args.reset_every == 0
    if fresh:
        ht.table.fill(0)
        ht.counts.fill(0)
        ht.names, ht.hashesperid = [], np.zeros(0, np.uint32)
        tb = TableBuilder(ht, ex0, prefault=True)
        tb._in_step_default = tb._in_step
        random.seed(4242)
        stored_since_reset = 0
    t0 = time.perf_counter()
    nh, per_batch = job(tb, it)
    t1 = time.perf_counter()
    stored_since_reset += nh
    tots = c3_steps(args.c3_steps)
    t2 = time.perf_counter()
    d10 = one_file(one10)
    d300 = one_file(one300) if it % 10 == 0 else None
    t3 = time.perf_counter()
    t_job += t1 - t0
    t_c3 += t2 - t1
    t_one += t3 - t2
    # ---- every iteration against the first ----
    got = dict(per_batch=per_batch, c3=tots, d10=d10)
    if d300 is not None:
        got['d300'] = d300
    for k, v in got.items():
        if k not in want:
            want[k] = v
        elif want[k] != v:
            mism.append((it, k, str(v)[:80], str(want[k])[:80]))
    if int(ht.counts.astype(np.int64).sum()) != stored_since_reset or int(np.asarray(ht.hashesperid, np.int64).sum()) != stored_since_reset:
        mism.append((it, 'counts do not add up', int(ht.counts.astype(np.int64).sum()), stored_since_reset))
    if fresh:
        # a fresh table after one job: the same bytes every time (same clips, same order, same random seed)
        dg = hashlib.sha256(ht.table.tobytes() + ht.counts.tobytes()).hexdigest()[:16]
        if 'fresh_table' not in want:
            want['fresh_table'] = dg
        elif want['fresh_table'] != dg:
            mism.append((it, 'fresh table digest', dg, want['fresh_table']))
    if mism:
        logline('MISMATCH %r' % (mism[-1],))
        if len(mism) > 20:
            break
    if it % 50 == 0 or it == args.iters - 1:
        ps = [e.path_stats()['redone_total'] for e in exs]
        logline('iter %d ok: job %.1f ms (%d rows, %d overflow draws so far), c3 %.2f ms/step, one-file %.2f ms, dense redo %s, mismatches %d'
                % (it, (t1 - t0) * 1e3, nh, tb.overflow_events, (t2 - t1) * 1e3 / max(1, args.c3_steps), (t3 - t2) * 1e3, ps, len(mism)))
sync_all()
out = dict(tag=args.tag, torch=not args.no_torch, iters_done=it + 1, iters_asked=args.iters, mismatches=len(mism), first_mismatches=mism[:5],
           faults=0, runtime=info, env={k: v for k, v in os.environ.items() if k.startswith(('AMD_', 'GPU_', 'AFP_'))},
           build=_lib.load().afp_build_id().decode(), job_clips=args.clips, c3_clips=args.c3_clips,
           mean_ms=dict(job=round(t_job / (it + 1) * 1e3, 2), c3_step=round(t_c3 / (it + 1) / max(1, args.c3_steps) * 1e3, 3), one_file=round(t_one / (it + 1) * 1e3, 3)),
           wall_s=round(time.perf_counter() - T0, 1), digests={k: v for k, v in want.items() if isinstance(v, str)})
logline('DONE ' + json.dumps(out))
sys.exit(1 if mism else 0)
