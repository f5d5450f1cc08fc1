#!/usr/bin/env python
"""bench.py -- landmark-hash extraction throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W            (single GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (PCM resident in HBM -> sorted unique (time, hash) rows
resident in HBM) over one batch of synthetic clips.  Workload (per GPU, weak scaling):
  c3  1024 x 30 s clips, density 20, fanout 3, 1 shift   (BASELINE configs[2]; DEFAULT -- the
      single-GPU throughput/roofline configuration)
  c5  1024 x 30 s, density 70, fanout 10, 4 shifts        (configs[4])
  c4  12500 x 10 s per GPU (= 100k over 8 GPUs)           (configs[3])
  c2  1 x 300 s                                           (configs[1]; the two threshold passes run segment-parallel)
The headline line is the `--workload` (c3); at N=1 the SAME line also carries the objects `c5`, `c4_slice`
and `c2_single_clip`: every single-GPU BASELINE configuration measured by the same command, each with its
ms per step, per-kernel times, roofline and a bit-exact parity check against the CPU oracle.
Clips shard across ranks with no data-path collective; torch.distributed is
used only for the barrier, the max-over-ranks of the elapsed time and the AND of the per-rank parity checks.

Batches are kept in flight the way a bulk ingest would: a few contexts share a spectral-stage stream and a
scan-stage stream (afp_set_stage_streams), so batch i+1's STFT runs beside batch i's scan; every step is
still a complete pass and `ms_per_step_one_context` is the same K steps strictly back to back.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the longest of k_stft / k_scan),
measured with HIP events on the launch stream inside this script; `cpu_baseline` times the numpy oracle
(oracle/afp_oracle.py, a restatement of the reference's numpy/scipy path -- it skips the reference's per-row
lfilter calls and Python peak-list loops, so it is if anything FASTER than the reference itself; kind "port") or,
when AFP_REF_DIR names a tree of the reference's own sources, the reference itself (kind "reference") on a bounded
sample of the same clips on the host, one thread, and over one process per CPU the container may use (`cpu_baseline_allcores`);
`parity` holds the rows of the LAST TIMED step (Runner.measure: the unguarded kernels that were timed) against the
oracle for EVERY clip of the batch; a further pass with the near-tie guard on only counts `near_tie_units`.

Layout of this file: helpers (clip pool, oracle process pool, sensors) -> Runner.measure (the timed region of every
resident-PCM workload) -> c4_job (the timed region of the ingest job) -> roofline_obj -> one function per object of
the JSON line (headline, cpu_baseline_one_core, extra_workload, ragged_workload, analyzer_path, table_build,
match_queries, c2_single_clip, the N > 1 extras) -> main().
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import audfprint_amd                                    # noqa: E402
audfprint_amd.configure_runtime()                       # GPU_MAX_HW_QUEUES, before torch touches the runtime (audfprint_amd/_lib.py says why)

WORKLOADS = {
    'c3': dict(nclips=1024, secs=30.0, density=20.0, fanout=3, shifts=1,
               name='1024 x 30 s synthetic 11025 Hz mono clips per GPU, density 20, fanout 3 (BASELINE configs[2])'),
    'c5': dict(nclips=1024, secs=30.0, density=70.0, fanout=10, shifts=4,
               name='1024 x 30 s clips per GPU, density 70, fanout 10, shifts 4 (BASELINE configs[4])'),
    'c4': dict(nclips=12500, secs=10.0, density=20.0, fanout=3, shifts=1,
               name='12500 x 10 s clips per GPU (100k over 8 GPUs), density 20 (BASELINE configs[3])'),
    'c2': dict(nclips=1, secs=300.0, density=20.0, fanout=3, shifts=1,
               name='single 300 s clip, density 20 (BASELINE configs[1])'),
}
SR = 11025
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6290.0   # what a float4 copy kernel reaches on this chip (same guide): the practical ceiling
FP64_PEAK_TF = 78.6         # vector FP64 peak = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (datasheet figure)
N_SIMD = 1024
DEFAULT_CU_SPLIT = 0        # CUs of the scan / pairing stages in staged mode (0: all stages share all CUs)
# FP64 work per STFT frame: real 512-point FFT 12.8 k + log 8 k + magnitude / HPF / compares /
# threshold updates 4.2 k = 25 kFLOP; one unit (clip x shift) of N samples has 1 + N // 256 frames
FLOP_PER_FRAME = 25000.0


class _TableArrays(object):
    """The arrays / fields of the reference HashTable that HashTable.store / merge touch (hash_table.py:61-83)
    plus name_to_id (:325-344) -- a plain container for audfprint_amd.table.TableBuilder to fill; on a
    real installation this is the reference's own hash_table.HashTable object."""

    def __init__(self, hashbits=20, depth=100, maxtime=16384):
        self.hashbits, self.depth, self.maxtimebits = hashbits, depth, int(round(np.log2(maxtime)))
        self.table = np.zeros((2 ** hashbits, depth), dtype=np.uint32)
        self.counts = np.zeros(2 ** hashbits, dtype=np.int32)
        self.names = []
        self.hashesperid = np.zeros(0, np.uint32)
        self.dirty = True

    def name_to_id(self, name, add_if_missing=False):
        if name not in self.names:
            self.names.append(name)
            self.hashesperid = np.append(self.hashesperid, [0])
        return self.names.index(name)


class _IndexedNames(list):
    """A list whose `in` / index() go through a dict: OracleHashTable.name_to_id (hash_table.py:325-344 restated) searches the
    names twice per store -- quadratic over a 12 500-clip job.  Names of the bench are unique strings, never removed."""

    def __init__(self):
        list.__init__(self)
        self._pos = {}

    def append(self, x):
        self._pos.setdefault(x, len(self))
        list.append(self, x)

    def __contains__(self, x):
        return x in self._pos

    def index(self, x, *a):
        if x in self._pos:
            return self._pos[x]
        raise ValueError(x)


NT_EPS = 1e-11      # near-tie epsilon of the GUARDED parity passes (afp_set_neartie_eps; the timed regions run with the library's default: off)


class guarded(object):
    """The near-tie guard on for the extractors given, off again afterwards.  The guard adds comparisons and changes no
    decision, so a guarded pass over the same batch on the same kernel path that marks nothing shows that every decision of
    the unguarded timed passes stood by a margin of NT_EPS."""

    def __init__(self, *exs):
        self.exs = exs

    def __enter__(self):
        for e in self.exs:
            e.set_neartie_eps(NT_EPS)

    def __exit__(self, *a):
        for e in self.exs:
            e.set_neartie_eps(0.0)


def _digest(h):
    return hashlib.sha256(np.ascontiguousarray(h, dtype='<i4').tobytes()).hexdigest()[:16]


def _cpu_worker(job):
    """One clip through the numpy oracle (runs in a spawned host process: the all-cores CPU baseline and the
    all-clips parity digest).  The clips live in a shared-memory block so nothing but an index crosses the pipe."""
    shm_name, shape, i, nsamp, kw = job
    from multiprocessing import shared_memory
    from oracle import afp_oracle as O
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        d = np.ndarray(shape, dtype=np.float32, buffer=shm.buf)[i, :nsamp].copy()
    finally:
        shm.close()
    h = cpu_rows_fn(O, kw)[0](d)              # (the reference itself when AFP_REF_DIR names its tree, else the oracle)
    return len(h), _digest(h)


def _cpu_worker_rows(job):
    """like _cpu_worker, returning the rows themselves (the table parity of c4_job stores them with the oracle's HashTable)"""
    shm_name, shape, i, nsamp, kw = job
    from multiprocessing import shared_memory
    from oracle import afp_oracle as O
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        d = np.ndarray(shape, dtype=np.float32, buffer=shm.buf)[i, :nsamp].copy()
    finally:
        shm.close()
    return cpu_rows_fn(O, kw)[0](d)


class OraclePool(object):
    """Spawned host processes running the CPU path (cpu_rows_fn: the reference under AFP_REF_DIR, else the oracle) over clips
    of a shared-memory pool (spawned, not forked: HIP is live in this process).  This is the reference's own --ncores scheme:
    file-sharded processes, audfprint.py:249."""

    def __init__(self, pool, nproc):
        self.kind = 'reference' if reference_tree() is not None else 'port'
        import multiprocessing as mp
        from multiprocessing import shared_memory
        self.shape = pool.shape
        self.shm = shared_memory.SharedMemory(create=True, size=pool.nbytes)
        np.ndarray(pool.shape, dtype=np.float32, buffer=self.shm.buf)[:] = pool
        self.nproc = nproc
        self.p = mp.get_context('spawn').Pool(nproc)
        self.p.map(_cpu_worker, [(self.shm.name, self.shape, 0, 2048, dict())] * nproc)     # start + import cost

    def run(self, idx, nsamp, kw):
        t0 = time.perf_counter()
        out = self.p.map(_cpu_worker, [(self.shm.name, self.shape, int(i), int(nsamp), kw) for i in idx], chunksize=2)
        return out, time.perf_counter() - t0

    def run_timed(self, idx, nsamp, kw, clips_per_proc=24):
        """The all-cores THROUGHPUT measurement: every process gets `clips_per_proc` clips (cycling over idx) in ONE task, so
        that what is timed is extraction on every core at once and not the dispatch of a few short tasks through one pipe
        (1024 clips over 256 processes are 4 clips = 80 ms each: r06 measured 0.62 M hashes/s that way, 6 x one core)."""
        idx = [int(i) for i in idx]
        jobs = [(self.shm.name, self.shape, idx[(w * clips_per_proc + k) % len(idx)], int(nsamp), kw)
                for w in range(self.nproc) for k in range(clips_per_proc)]
        t0 = time.perf_counter()
        out = self.p.map(_cpu_worker, jobs, chunksize=clips_per_proc)
        return out, time.perf_counter() - t0

    def rows(self, idx, nsamp, kw):
        return self.p.map(_cpu_worker_rows, [(self.shm.name, self.shape, int(i), int(nsamp), kw) for i in idx], chunksize=2)

    def run_var(self, idx, nsamps, kw):
        """like run, one length per clip"""
        t0 = time.perf_counter()
        out = self.p.map(_cpu_worker, [(self.shm.name, self.shape, int(i), int(n), kw) for i, n in zip(idx, nsamps)], chunksize=2)
        return out, time.perf_counter() - t0

    def close(self):
        try:
            self.p.close()
            self.p.join()
        finally:
            self.shm.close()
            self.shm.unlink()


def synth_pool(npool, nsamp, seed0):
    """SURVEY.md §8c/§8d recipe: clip i = RandomState(seed0 + i): white Gaussian sigma 0.1, clipped,
    int16-quantised, /32768 -> float32."""
    out = np.empty((npool, nsamp), dtype=np.float32)
    for i in range(npool):
        rng = np.random.RandomState(seed0 + i)
        x = rng.randn(nsamp) * 0.1
        pcm = np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
        out[i] = pcm.astype(np.float32) / np.float32(32768)
    return out


def frames_of(nsamp, shifts):
    offs = [0] if shifts < 2 else [int(s / shifts * 256) for s in range(shifts)]
    return sum(1 + (nsamp - o) // 256 for o in offs if nsamp - o > 0)


class _RawStream(object):
    """A hipStream_t created by the library (CU-masked); quacks like torch.cuda.Stream where bench.py needs it."""

    def __init__(self, raw):
        self.cuda_stream = raw


class GpuSensors(object):
    """Package power and shader clock of one GPU from its hwmon sysfs files, sampled by a thread while the pipeline runs
    (the result is power / clock sensitive -- show what the box saw).  ok = False when the files are absent."""

    def __init__(self, torch, dev):
        import glob
        self.files = {}
        try:
            pr = torch.cuda.get_device_properties(dev)
            addr = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for hw in glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % addr):
                for key, names in (('power_uw', ('power1_average', 'power1_input')), ('sclk_hz', ('freq1_input',))):
                    for nm in names:
                        fp = os.path.join(hw, nm)
                        if key not in self.files and os.path.exists(fp):
                            self.files[key] = fp
        except Exception:       # noqa: BLE001
            self.files = {}
        self.ok = bool(self.files)
        self.samples = {k: [] for k in self.files}
        self._stop = False
        self._th = None

    def _run(self):
        while not self._stop:
            for k, fp in self.files.items():
                try:
                    with open(fp) as f:
                        self.samples[k].append(float(f.read().strip()))
                except (OSError, ValueError):
                    pass
            time.sleep(0.004)

    def start(self):
        import threading
        self._stop = False
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th is not None:
            self._th.join(1.0)
        out = dict(source='hwmon sysfs of the GPU, sampled every ~4 ms while the pipeline ran')
        p, c = self.samples.get('power_uw', []), self.samples.get('sclk_hz', [])
        if p:
            out.update(power_w_mean=round(sum(p) / len(p) / 1e6, 1), power_w_max=round(max(p) / 1e6, 1), samples=len(p))
        if c:
            out.update(sclk_mhz_mean=round(sum(c) / len(c) / 1e6, 1), sclk_mhz_min=round(min(c) / 1e6, 1))
        return out


class Runner(object):
    """The contexts (batches in flight) of one GPU and the measurement of one workload on them."""

    def cu_split(self):
        s = self.args.cu_split
        return DEFAULT_CU_SPLIT if s < 0 else s

    def __init__(self, args, torch, dev, local_rank, dist):
        from audfprint_amd.batch import Extractor
        self.args, self.torch, self.dev, self.dist = args, torch, dev, dist
        self.ex = Extractor.get(local_rank)
        self.local_rank = local_rank
        self.Extractor = Extractor
        self.extra = []                    # further contexts, created on demand
        self.spectral = None
        self.stage_sets = []

    def contexts(self, n, staged):
        torch, dev, args = self.torch, self.dev, self.args
        while len(self.extra) < n - 1:
            self.extra.append(self.Extractor(self.local_rank))
        exs = [self.ex] + self.extra[:n - 1]
        if staged and n > 1:
            if self.spectral is None:
                # one spectral-stage stream shared by all contexts; `--scan-streams` scan(/pair)-stage stream sets,
                # contexts take them round-robin (1: scans strictly one after another)
                split = self.cu_split()
                if split > 0:
                    # CU-partitioned pipeline: the scan / pairing stages own `split` compute units, the spectral stage
                    # the others (streams created through the library: hipExtStreamCreateWithCUMask)
                    from audfprint_amd.batch import cu_range_stream
                    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
                    mk = lambda first, n: _RawStream(cu_range_stream(self.local_rank, first, n))
                    self.spectral = mk(split, ncu - split)
                    for _ in range(max(1, args.scan_streams)):
                        ss = [self.spectral, mk(0, split)]
                        if args.stages >= 3:
                            ss.append(mk(0, split))
                        self.stage_sets.append(ss)
                else:
                    self.spectral = torch.cuda.Stream(device=dev)
                    for _ in range(max(1, args.scan_streams)):
                        ss = [self.spectral, torch.cuda.Stream(device=dev, priority=args.scan_prio)]
                        if args.stages >= 3:
                            ss.append(torch.cuda.Stream(device=dev, priority=args.scan_prio))
                        self.stage_sets.append(ss)
            for i, e in enumerate(exs):
                e.set_stage_streams(*[s_.cuda_stream for s_ in self.stage_sets[i % len(self.stage_sets)]])
        else:
            for e in exs:
                e.set_stage_streams(None, None, None)
        return exs

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, wl, d_pcm, offsets, steps, warmup, overlap=True, staged=-1, inflight=0, keep_result=True):
        """W untimed warmup steps, then exactly K timed steps bracketed by barrier + synchronize; returns a dict.
        `timed_res` = the BatchResult of the last timed step (fetched right after the timed region, outside it)."""
        args = self.args
        if staged < 0:
            staged = 1          # (r02: since k_stft needs 25 KB of LDS the staged arrangement also wins on the multi-shift C5)
        if inflight <= 0:
            inflight = 4 if staged else 3          # (measured r02: three unstaged contexts beat two on C5)
        exs = self.contexts(inflight if overlap else 1, staged)
        ex = self.ex
        for e in exs:
            e.set_params(density=wl['density'], maxpairsperpeak=wl['fanout'], shifts=wl['shifts'])
        ptr = d_pcm.data_ptr()

        def step():
            ex.extract_device(ptr, offsets, want_hashes=True, want_peaks=False)
            return ex.counts()[0]          # synchronises: results are resident in HBM

        def run_steps(n):
            """n complete passes of the hot path; at most len(exs) batches in flight."""
            nh_ = 0
            fl = []
            for i in range(n):
                e = exs[i % len(exs)]
                if len(fl) == len(exs):
                    nh_ = fl.pop(0).counts()[0]          # waits for that batch: results resident in HBM
                e.extract_device(ptr, offsets, want_hashes=True, want_peaks=False)
                fl.append(e)
            for e in fl:
                nh_ = e.counts()[0]
            return nh_

        # prime every context once (workspace allocation, descriptor upload, output sizing) -- setup, not a step;
        # then the W untimed warmup steps of the contract
        for e in exs:
            e.extract_device(ptr, offsets, want_hashes=True, want_peaks=False)
            e.counts()
        nh = run_steps(warmup)
        self.barrier()
        t0 = time.perf_counter()
        nh = run_steps(steps)
        self.barrier()
        elapsed = time.perf_counter() - t0
        # the rows the LAST TIMED step left in HBM, copied out before anything else touches that context: this -- the output
        # of the kernels that were timed, guard off -- is what the parity objects hold against the oracle
        timed_res = None
        if keep_result and steps > 0:
            last = exs[(steps - 1) % len(exs)]
            timed_res = last.fetch(len(offsets) - 1, True, False)
            timed_res.path = last.path_stats()
        # the same K steps strictly back to back on one context (no overlap between batches)
        self.barrier()
        ts0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.barrier()
        serial_ms = (time.perf_counter() - ts0) / steps * 1e3
        # shader clock actually held while the pipeline runs (s_memtime against the constant 100 MHz s_memrealtime)
        mhz = None
        try:
            ex_probe = exs[-1]
            ex_probe.clock_probe_start()
            run_steps(max(4, min(steps, 10)))
            mhz = ex_probe.clock_probe_stop()
        except Exception:
            mhz = None
        # package power / sclk the box reports while the same pipeline runs (~0.3 s of steps, outside the timed region)
        power = None
        try:
            sens = GpuSensors(self.torch, self.dev)
            if sens.ok:
                n_p = max(8, min(400, int(0.3 / max(1e-4, elapsed / max(1, steps)))))
                sens.start()
                run_steps(n_p)
                power = sens.stop()
        except Exception:       # noqa: BLE001
            power = None
        # ---- per-kernel timing (HIP events on the launch stream), after the timed region ----------
        ex.set_stage_streams(None, None, None)
        ex.set_timing(True)
        ex.reset_timings()
        for _ in range(max(3, min(steps, 10))):
            step()
        tm = ex.timings()
        ex.set_timing(False)
        kern_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in tm.items()}
        return dict(elapsed=elapsed, nh=nh, serial_ms=serial_ms, kern_ms=kern_ms, nctx=len(exs),
                    staged=(len(self.stage_sets[0]) if (staged and len(exs) > 1) else 0), mhz=mhz, power=power,
                    timed_res=timed_res)


def numa_of_gpu(torch, dev):
    """(numa node, cpu list) of the GPU `dev` hangs off, from sysfs (PCI address of the device); (-1, []) if unknown."""
    try:
        pr = torch.cuda.get_device_properties(dev)
        addr = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open('/sys/bus/pci/devices/%s/numa_node' % addr) as f:
            node = int(f.read().strip())
        if node < 0:
            return node, []
        with open('/sys/devices/system/node/node%d/cpulist' % node) as f:
            cpus = []
            for part in f.read().strip().split(','):
                a, _, b = part.partition('-')
                cpus.extend(range(int(a), int(b or a) + 1))
        return node, cpus
    except Exception:       # noqa: BLE001
        return -1, []


def c4_job(R, torch, pool, npool, rank, nclips_job, batch, nctx, O, opool, parity_batches=2, seed=0, whole_job_parity=False):
    """BASELINE configs[3] AS THE JOB IT NAMES, on one GPU's slice: `new -> fpdbase` (audfprint.py:173-186 per file:
    Analyzer.ingest -> wavfile2hashes -> HashTable.store, hash_table.py:91-138) over nclips_job x 10 s clips.
    Raw s16 PCM (what ffmpeg pipes) sits in PINNED host memory; batches of `batch` clips go through `nctx` staged contexts
    (Extractor.submit: the upload of batch i+1 runs under the kernels of batch i), each batch's rows stay in HBM and are
    stored into ONE device table in clip order (TableBuilder.store_batch(src=ctx) = afp_table_store_device, the overflow
    draws replayed from Python's own generator state), and finalize() copies the table into the HashTable's host arrays.
    Timed: first submit .. host arrays complete.  Returns (report dict, TableBuilder, hashtable) -- the table stays on the
    device for the cross-rank merge at N > 1."""
    setup_err = None
    try:
        import random
        from audfprint_amd.table import TableBuilder
        w = dict(WORKLOADS['c4'])
        ns = int(round(w['secs'] * SR))
        if pool.shape[1] < ns:
            raise ValueError('c4_job needs pool clips of at least %.0f s' % w['secs'])
        h16 = np.round(pool[:, :ns] * 32768).astype(np.int16)            # exact: the pool is int16 / 32768 (audio_read.buf_to_float)
        nb = (nclips_job + batch - 1) // batch
        pin = torch.empty((nclips_job, ns), dtype=torch.int16).pin_memory()
        pin_np = pin.numpy()
        for lo in range(0, nclips_job, npool):
            hi = min(nclips_job, lo + npool)
            pin_np[lo:hi] = h16[:hi - lo]
        flat = pin_np.reshape(-1)
        names = ['r%dclip%06d' % (rank, i) for i in range(nclips_job)]
        exs = R.contexts(nctx, 1)
        for e in exs:
            e.set_params(density=w['density'], maxpairsperpeak=w['fanout'], shifts=w['shifts'])

    except Exception as e:       # noqa: BLE001   (raised again BEHIND the barrier every rank must reach)
        setup_err = e

    def run(nbatches, rseed):
        ht = _TableArrays(hashbits=20, depth=100)        # (np.zeros: untouched pages, like the reference's fresh HashTable)
        random.seed(rseed)
        pend, nh, wait_s = [], 0, 0.0
        nt_units = [0]
        torch.cuda.synchronize()                 # (local: no collective inside the job -- a rank that fails must not strand the others)
        t0 = time.perf_counter()
        # the builder is part of the job: creating it zeroes the device table and -- prefault=True -- starts the
        # background population of the host table's pages, which the reference's np.zeros table pays inside its stores
        tb = TableBuilder(ht, R.ex, prefault=True)       # (the host waits ~7 ms for the first batch: the fresh table's pages are populated meanwhile)
        t_created = time.perf_counter() - t0

        marks = [] if os.environ.get('AFP_C4_TRACE') else None     # host-side timeline of the job (ms since its start)

        def mark(what):
            if marks is not None:
                marks.append((what, round((time.perf_counter() - t0) * 1e3, 3)))

        def retire():
            e, lo, hi = pend.pop(0)
            tw = time.perf_counter()
            off = e.fetch_offsets(hi - lo)                           # waits for that batch; the rows stay in HBM
            tw = time.perf_counter() - tw
            nt_units[0] += e.path_stats()['near_tie_units']
            mark('fetched %d' % (lo // batch))
            tb.store_batch(names[lo:hi], offsets=off, src=e)
            mark('stored %d' % (lo // batch))
            return int(off[-1]), tw
        for b in range(nbatches):
            lo, hi = b * batch, min(nclips_job, (b + 1) * batch)
            e = exs[b % len(exs)]
            if len(pend) == len(exs):
                n_, tw = retire()
                nh += n_
                wait_s += tw
            e.submit(flat[lo * ns:hi * ns], np.arange(hi - lo + 1, dtype=np.int64) * ns)
            mark('submitted %d' % b)
            pend.append((e, lo, hi))
        while pend:
            n_, tw = retire()
            nh += n_
            wait_s += tw
        t1 = time.perf_counter()
        tb.finalize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if marks is not None:
            sys.stderr.write('c4_job host timeline (ms): %s | stores done %.3f | arrays complete %.3f\n' % (marks, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
        return dict(tb=tb, ht=ht, nh=nh, wait_s=wait_s, t_store_done=t1 - t0, t_total=t2 - t0, nclips=min(nclips_job, nbatches * batch),
                    near_tie_units=nt_units[0], t_created=t_created)

    # ---- parity first (also the warm-up of every context): the job's own code path on its first `parity_batches` batches,
    #      table / counts / names / hashesperid against OracleHashTable.store of the oracle's rows, same order, same seed ----
    par = None
    pb = 0
    nt_prefix = None
    oracle_rows = None
    try:
        if setup_err is not None:
            raise setup_err
        pb = min(parity_batches, nb)
        with guarded(*exs):                       # (the prefix holds every distinct clip of the job: pool size <= its clips)
            rp = run(pb, seed)
        nt_prefix = int(rp['near_tie_units'])
        if O is not None:
            ncl = rp['nclips']
            kw = dict(density=w['density'], maxpairsperpeak=w['fanout'], shifts=w['shifts'])
            tq = time.perf_counter()
            # (the rows of EVERY distinct clip of the job, once: the whole-job check behind the timed run uses them again)
            distinct = list(range(min(npool, nclips_job if whole_job_parity else ncl)))
            rows = opool.rows(distinct, ns, kw) if opool is not None else [O.extract(pool[i, :ns], O.Params(**kw))[1] for i in distinct]
            oracle_rows = rows
            ref = O.OracleHashTable(hashbits=20, depth=100)
            rr = random.Random(seed)
            for i in range(ncl):
                ref.store(names[i], rows[i % npool], rr)
            tq = time.perf_counter() - tq
            g = rp['ht']
            ok = (np.array_equal(g.table, ref.table) and np.array_equal(g.counts, ref.counts) and g.names == ref.names and
                  np.array_equal(np.asarray(g.hashesperid, np.int64), np.asarray(ref.hashesperid, np.int64)))
            par = dict(clips_checked=ncl, bit_exact=bool(ok), rows=int(rp['nh']), overflow_draws=int(rp['tb'].overflow_events),
                       buckets_over_depth=int(np.sum(ref.counts > 100)),
                       how='the job itself on its first %d batches (%d clips, same contexts, same batch size): table, counts, names and '
                           'hashesperid equal OracleHashTable.store of the oracle\'s rows clip by clip with random.seed(%d) '
                           '(%.1f s of oracle work)' % (pb, ncl, seed, tq))
        del rp
        # warm-up of what the parity run did not touch: the contexts beyond its batches (their first submit allocates the
        # PCM stage and the workspace -- 16 ms inside the r04 job when context 2 met its first batch in the timed run) and
        # the first re-growth of the row-sized table buffers (a hipFree = a device-wide wait behind the queued uploads: 8 ms)
        if nb > pb:
            rw = run(min(nb, len(exs) + 1), seed)
            del rw
    except Exception as e:       # noqa: BLE001   (reported; the timed job still runs, and every rank still reaches the barrier below)
        par = dict(bit_exact=False, error=repr(e))
    R.barrier()                                   # ranks start the timed job together; the ONLY collective of this function
    if setup_err is not None:
        raise setup_err
    # ---- the timed job --------------------------------------------------------------------------------------------
    r = run(nb, seed)
    tb, ht = r['tb'], r['ht']
    audio = nclips_job * w['secs']
    tot_cnt = int(ht.counts.astype(np.int64).sum())
    sec = tb.seconds
    # what the stages cost on their own (one batch, nothing else running): the upload, the kernels on resident PCM
    lo, hi = 0, min(batch, nclips_job)
    d_one = torch.empty((hi - lo) * ns, dtype=torch.int16, device=R.dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(3):
        d_one.copy_(pin.view(-1)[lo * ns:hi * ns], non_blocking=True)
    ev1.record()
    torch.cuda.synchronize()
    h2d_ms = ev0.elapsed_time(ev1) / 3
    off1 = np.arange(hi - lo + 1, dtype=np.int64) * ns
    e = R.ex
    e.set_stage_streams(None, None, None)
    for _ in range(2):
        e.extract_device(d_one.data_ptr(), off1, want_hashes=True, want_peaks=False, s16=True)
        e.counts()
    tk = time.perf_counter()
    for _ in range(3):
        e.extract_device(d_one.data_ptr(), off1, want_hashes=True, want_peaks=False, s16=True)
        e.counts()
    kern_ms = (time.perf_counter() - tk) / 3 * 1e3
    bytes_job = float(nclips_job) * ns * 2
    out = dict(workload='%d x %.0f s s16 clips: pinned host PCM -> pipelined H2D (%d staged contexts, batches of %d) -> extract -> '
                        'afp_table_store_device from the resident rows -> finalize() into host HashTable arrays  [BASELINE configs[3], '
                        'one GPU\'s slice of 100k clips over 8 GPUs]' % (nclips_job, w['secs'], len(exs), batch),
               clips=nclips_job, batches=nb, batch_clips=batch, contexts=len(exs), hashes=int(r['nh']),
               job_ms=round(r['t_total'] * 1e3, 2), hashes_per_s=round(r['nh'] / r['t_total'], 1),
               audio_sec_per_sec=round(audio / r['t_total'], 1),
               pcie_gb_per_s_over_job=round(bytes_job / r['t_total'] / 1e9, 2),
               pcm_bytes=int(bytes_job),
               runtime=dict(GPU_MAX_HW_QUEUES=os.environ.get('GPU_MAX_HW_QUEUES'), upload_stream=os.environ.get('AFP_UPLOAD_STREAM', '1'),
                            download_threads=os.environ.get('AFP_DL_THREADS', '8'),
                            warmup='parity run on the first %d batches, then %d batches through every context' % (pb, min(nb, len(exs) + 1) if nb > pb else 0)),
               stages_ms=dict(table_builder_created=round(r['t_created'] * 1e3, 3),
                              until_last_store=round(r['t_store_done'] * 1e3, 2),
                              waiting_for_batches=round(r['wait_s'] * 1e3, 2),
                              table_store_kernels=round(sec['store'] * 1e3, 2),
                              overflow_replay=round(sec['replay'] * 1e3, 2),
                              download_to_host_arrays=round(sec['download'] * 1e3, 2),
                              note='host-side wall time of the pipelined job, measured from BEFORE the TableBuilder is created '
                                   '(device table zeroed, host pages\' background prefault started): waiting = blocked until a '
                                   'batch\'s upload + kernels had finished; store / replay / download block the host'),
               one_batch_alone_ms=dict(h2d=round(h2d_ms, 3), h2d_gb_per_s=round((hi - lo) * ns * 2 / (h2d_ms * 1e-3) / 1e9, 1),
                                       kernels_resident_s16=round(kern_ms, 3), batches=nb,
                                       h2d_sum_over_job=round(h2d_ms * nclips_job / (hi - lo), 2),
                                       kernels_sum_over_job=round(kern_ms * nclips_job / (hi - lo), 2)),
               overflow_draws=int(tb.overflow_events), table_total_count=tot_cnt, ids=len(ht.names),
               invariants=dict(counts_add_up=bool(tot_cnt == int(r['nh'])), every_clip_has_an_id=bool(len(ht.names) == nclips_job),
                               hashesperid_adds_up=bool(int(np.asarray(ht.hashesperid, np.int64).sum()) == int(r['nh']))),
               table_bytes=int((1 << 20) * 100 * 4 + (1 << 20) * 4))
    out['table_bytes_downloaded'] = int(tb.bytes_downloaded)
    out['near_tie_units'] = nt_prefix
    out['near_tie_how'] = 'guarded pass over the first %d batches (every distinct clip of the job); the timed job runs unguarded' % pb
    if par is not None and par.get('bit_exact') and whole_job_parity and oracle_rows is not None:
        # ---- the TIMED job's table against OracleHashTable.store over ALL its clips, same order, same seed: every overflow
        #      draw of the job, the full-bucket regime of the late batches included (hash_table.py:91-138) ----
        try:
            tq = time.perf_counter()
            ref = O.OracleHashTable(hashbits=20, depth=100)
            ref.names = _IndexedNames()                      # (list.index over 12 500 names per store: 3 s of nothing but string compares)
            rr = random.Random(seed)
            for i in range(nclips_job):
                ref.store_fast(names[i], oracle_rows[i % npool], rr)
            tq = time.perf_counter() - tq
            ok = (np.array_equal(ht.table, ref.table) and np.array_equal(ht.counts, ref.counts) and list(ht.names) == list(ref.names) and
                  np.array_equal(np.asarray(ht.hashesperid, np.int64), np.asarray(ref.hashesperid, np.int64)))
            par = dict(par, prefix=dict(clips_checked=par['clips_checked'], overflow_draws=par['overflow_draws'], how=par['how']),
                       clips_checked=int(nclips_job), bit_exact=bool(ok), rows=int(r['nh']), overflow_draws=int(tb.overflow_events),
                       buckets_over_depth=int(np.sum(ref.counts > 100)),
                       how='the TIMED job\'s host arrays (table, counts, names, hashesperid) equal OracleHashTable.store_fast (store()\'s loop batched per clip, held against store() row for row in tests/) of the oracle\'s rows '
                           'over all %d clips in job order with random.seed(%d): all %d overflow draws (%.1f s of oracle work); the '
                           'first %d batches were checked the same way before the timed run' % (nclips_job, seed, int(tb.overflow_events), tq, pb))
        except Exception as e:       # noqa: BLE001
            par = dict(par, bit_exact=False, error='whole-job check: ' + repr(e))
    if par is not None:
        out['parity'] = par
    del d_one, pin
    return out, tb, ht


def host_pipelined(R, torch, pool, npool, nsamp, wl, nh_clips, tags=('float32', 's16'), nrep=12):
    """PCIe-inclusive rate with the upload of batch i + 1 under the kernels of batch i: pinned host buffers (so the copy is
    asynchronous), three staged contexts, rows copied back to host arrays for every batch."""
    h_pcm = np.ascontiguousarray(pool[np.arange(nh_clips) % npool, :nsamp].reshape(-1))
    h_off = np.arange(nh_clips + 1, dtype=np.int64) * nsamp
    arrs = dict(float32=h_pcm)
    if 's16' in tags:
        arrs['s16'] = np.round(h_pcm * 32768).astype(np.int16)
    exs = R.contexts(3, 1)
    for e in exs:
        e.set_params(density=wl['density'], maxpairsperpeak=wl['fanout'], shifts=wl['shifts'])
    pip = {}
    for tag in tags:
        arr = arrs[tag]
        pins = [torch.from_numpy(arr.copy()).pin_memory().numpy() for _ in exs]
        for e, pn in zip(exs, pins):                      # prime (workspace, staging buffer, output sizing)
            e.submit(pn, h_off)
            e.fetch(nh_clips, True, False)
        fl = []
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        nhh = 0
        for i in range(nrep):
            k = i % len(exs)
            if len(fl) == len(exs):
                nhh = len(fl.pop(0).fetch(nh_clips, True, False).hashes)
            exs[k].submit(pins[k], h_off)
            fl.append(exs[k])
        for e in fl:
            nhh = len(e.fetch(nh_clips, True, False).hashes)
        th = (time.perf_counter() - th0) / nrep
        pip[tag] = dict(ms_per_batch=round(th * 1e3, 3), clips=nh_clips, hashes_per_s=round(nhh / th, 1),
                        audio_sec_per_sec=round(nh_clips * wl['secs'] / th, 1),
                        pcie_gb_per_s=round(arr.nbytes / th / 1e9, 1))
    pip['how'] = '3 staged contexts, pinned host PCM, H2D of batch i+1 under the kernels of batch i, rows fetched to host'
    R.contexts(1, 0)
    return pip


def load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def roofline_obj(key, wl, nclips, nsamp, nh, ms_per_step, kern_ms, mhz, build_id=None):
    """ALGORITHMIC bytes: float32 PCM read once + (N,2) int32 rows written once, over the dominant
    kernel's mean duration; next to it the roofline that actually binds this path -- FP64 vector issue."""
    alg_bytes = 4.0 * nclips * nsamp + 8.0 * float(nh)
    dom = max(((k, v) for k, v in kern_ms.items() if not k.startswith('pipeline')), key=lambda kv: kv[1])
    achieved = alg_bytes / (dom[1] * 1e-3) / 1e9 if dom[1] > 0 else 0.0
    tj = load_json(os.path.join(ROOT, 'profiles', 'traffic.json'))
    traffic = tj.get(key, {})
    pmc = load_json(os.path.join(ROOT, 'profiles', 'pmc.json')).get(key, {})
    # the committed counters belong to ONE build of the library: reported only when that build is the one running
    prof_id = tj.get(key + '_build_id')
    profiled = bool(traffic) or bool(pmc)
    stale = profiled and build_id is not None and (prof_id != build_id or pmc.get('build_id') != build_id)
    if stale:
        traffic, pmc = {}, {}
    flops = FLOP_PER_FRAME * nclips * frames_of(nsamp, wl['shifts'])
    tf = flops / (ms_per_step * 1e-3) / 1e12
    out = dict(bound='hbm', kernel=dom[0], achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit='GB/s',
               frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic.get(dom[0]),
               frac_of_measured_copy_ceiling=round(achieved / HBM_COPY_CEILING_GBS, 5),      # also against the 6.29 TB/s a float4 copy reaches
               alg_bytes_per_launch=alg_bytes, kernel_ms=round(dom[1], 4),
               whole_step_frac=round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
               kernels_ms={k: round(v, 4) for k, v in kern_ms.items()},
               fp64=dict(flop_per_step=flops, flop_per_frame=FLOP_PER_FRAME, achieved=round(tf, 3), peak=FP64_PEAK_TF,
                         unit='TFLOP/s', frac=round(tf / FP64_PEAK_TF, 4), over='whole step'))
    out['profile_build_id'] = prof_id
    if not profiled:
        out['no_counter_profile'] = 'no PMC pass is committed for this workload (profiles/traffic.json has no key %r): traffic is null' % key
    if stale:
        out['stale_profile'] = ('profiles/traffic.json / pmc.json were taken from build %s, this library is %s: traffic and '
                                'valu_issue withheld' % (prof_id, build_id))
    if traffic:
        tot = float(sum(v for k, v in traffic.items() if isinstance(v, (int, float))))
        out['traffic_step_total'] = tot
        out['traffic_over_algorithmic'] = round(tot / alg_bytes, 3)
        out['hbm_moved_gbs_step'] = round(tot / (ms_per_step * 1e-3) / 1e9, 1)
    if pmc.get('valu_quad_cycles'):
        clk = (mhz or 2400.0) * 1e6
        busy = 4.0 * float(pmc['valu_quad_cycles'])          # SQ_ACTIVE_INST_VALU counts quad-cycles
        vfrac = busy / N_SIMD / (ms_per_step * 1e-3 * clk)
        out['valu_issue'] = dict(valu_busy_cycles_per_step=busy, simds=N_SIMD, shader_mhz=mhz,
                                 frac=round(vfrac, 4), source=pmc.get('source'))
        # which ceiling actually binds: the VALU issue slots of the 1024 SIMDs at the clock the package power limit leaves
        # (frac above) against the HBM bytes really moved (hbm_moved / 8 TB/s).  `achieved / peak / frac` stay the contract's
        # algorithmic-bytes-over-HBM-peak figure whatever binds.
        moved_frac = out.get('hbm_moved_gbs_step', 0.0) / HBM_PEAK_GBS
        # `bound` keeps the contract's vocabulary ("hbm" | "mfma": the roofline `achieved` / `peak` are priced against);
        # `limited_by` says which resource actually runs out first
        out['limited_by'] = 'valu_issue' if vfrac > moved_frac else 'hbm'
        if vfrac > moved_frac:
            out['bound_note'] = ('VALU issue %.2f of the SIMDs\' cycles at the measured %.0f MHz vs %.2f of HBM peak moved: this FP64 path is '
                                 'instruction-issue bound under the package power limit; frac (algorithmic bytes / HBM peak) is kept as the '
                                 'contract defines it' % (vfrac, mhz or 2400.0, moved_frac))
        out['hbm_frac'] = out['frac']
    return out


def gpu_digests(res, idx):
    return [(int(res.hash_offsets[i + 1] - res.hash_offsets[i]), _digest(res.clip_hashes(i))) for i in idx]


def same_rows(a, b):
    """two BatchResults hold the same rows for the same clips"""
    return bool(np.array_equal(a.hash_offsets, b.hash_offsets) and np.array_equal(a.hashes, b.hashes))


# ======================================================================================================================
#  The CPU path timed beside the GPU
# ======================================================================================================================
def cgroup_cpu_limit():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited /
    unknown: os.cpu_count() counts the host's CPUs, not what the container is allowed to burn."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p_ = f.read().split()[:2]
        return None if q == 'max' else round(float(q) / float(p_), 2)
    except Exception:       # noqa: BLE001
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            q = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            p_ = float(f.read())
        return None if q <= 0 else round(q / p_, 2)
    except Exception:       # noqa: BLE001
        return None


def effective_cpus():
    """CPUs this process can actually keep busy: os.cpu_count(), narrowed by its affinity mask and by the container's cgroup
    quota (the GPU boxes of this pool show 256 CPUs and a quota of 16: 256 oracle processes then share 16 CPUs' worth of time
    and the "all cores" run measures the throttle -- 0.68 M hashes/s in 17.7 s, r06)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:       # noqa: BLE001
        pass
    q = cgroup_cpu_limit()
    if q is not None:
        n = min(n, max(1, int(q + 0.999)))
    return max(1, n)


def reference_tree():
    """The directory of the reference's own sources when the caller points at one with AFP_REF_DIR (never looked for
    anywhere else: the GPU box has none), else None."""
    d = os.environ.get('AFP_REF_DIR', '').strip()
    return d if d and os.path.isfile(os.path.join(d, 'audfprint_analyze.py')) else None


_REF_MOD = {}


def reference_rows(ref_dir, d, kw):
    """One clip through the REFERENCE ITSELF (audfprint_analyze.py:255-343, 81-96, the shifts loop :369-377 and the
    unique / sort of wavfile2hashes :404-422), imported unchanged from `ref_dir`."""
    A = _REF_MOD.get(ref_dir)
    if A is None:
        import importlib.util
        sys.path.insert(0, ref_dir)                       # (its own `import stft`, `import audio_read`)
        try:
            spec = importlib.util.spec_from_file_location('_afp_reference_analyze', os.path.join(ref_dir, 'audfprint_analyze.py'))
            A = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(A)
        finally:
            sys.path.remove(ref_dir)
        _REF_MOD[ref_dir] = A
    an = A.Analyzer(kw.get('density', 20.0))
    an.maxpairsperpeak, an.shifts = kw.get('maxpairsperpeak', 3), kw.get('shifts', 1)
    nsh = 1 if (an.shifts is None or an.shifts < 2) else int(an.shifts)
    offs = [0] if nsh == 1 else [int(sh / an.shifts * an.n_hop) for sh in range(nsh)]
    hs = [A.landmarks2hashes(an.peaks2landmarks(an.find_peaks(d[o:], SR))) for o in offs]
    h = np.concatenate(hs) if hs else np.zeros((0, 2), np.int32)
    if not len(h):
        return np.zeros((0, 2), np.int32)
    k = np.sort(np.unique((h[:, 0].astype(np.uint64) << np.uint64(32)) + h[:, 1].astype(np.uint64)))
    return np.stack([(k >> np.uint64(32)).astype(np.int32), (k & np.uint64(0xffffffff)).astype(np.int32)], axis=1)


def cpu_rows_fn(O, kw):
    """(f(d) -> rows, kind): the reference itself when AFP_REF_DIR names its tree (kind "reference"), else the oracle's
    restatement of it (kind "port")."""
    ref = reference_tree()
    if ref is not None:
        return (lambda d: reference_rows(ref, d, kw)), 'reference'
    prm = O.Params(**kw)
    return (lambda d: O.extract(d, prm)[1]), 'port'


# ======================================================================================================================
#  One process of the benchmark
# ======================================================================================================================
class Bench(object):
    """What the workload functions share: the arguments, this rank and its GPU, the Runner (contexts), the pool of synthetic
    clips, the oracle module (None with --no-cpu) and the pool of host processes running it (N = 1, opened on demand)."""

    def __init__(self, args):
        self.args = args
        self.O = None
        self.opool = None
        self.line_guard = None

    def kw(self, w):
        return dict(density=w['density'], maxpairsperpeak=w['fanout'], shifts=w['shifts'])

    def resident(self, nclips_, nsamp_):
        """`nclips_` clips of `nsamp_` samples resident in HBM (pool clips tiled), with their offsets."""
        torch = self.torch
        reps = (nclips_ + self.npool - 1) // self.npool
        d_pool = torch.from_numpy(np.ascontiguousarray(self.pool[:, :nsamp_])).to(self.dev)
        d = d_pool.repeat(reps, 1)[:nclips_].contiguous().view(-1)
        return d, np.arange(nclips_ + 1, dtype=np.int64) * nsamp_

    def gather(self, obj):
        if self.dist is None or self.world == 1:
            return [obj]
        lst = [None] * self.world
        self.dist.all_gather_object(lst, obj)
        return lst

    def emit(self, out):
        sys.stdout.flush()
        os.write(self.json_fd, (json.dumps(out) + '\n').encode())


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--nclips', type=int, default=0, help='override clips per GPU')
    ap.add_argument('--secs', type=float, default=0.0, help='override clip length')
    ap.add_argument('--pool', type=int, default=1024, help='distinct synthetic clips generated per GPU (tiled to nclips)')
    ap.add_argument('--cpu-sample', type=int, default=512, help='clips timed on the CPU path, one thread (rank 0, N=1)')
    ap.add_argument('--cpu-procs', type=int, default=0, help='host processes of the all-cores CPU baseline (0 = the CPUs this container may use: os.cpu_count() within affinity and cgroup quota)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline and the oracle parity checks')
    ap.add_argument('--no-cpu-all', action='store_true', help='skip the all-cores CPU baseline / all-clips parity extra')
    ap.add_argument('--no-c2', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the c5 / c4_slice objects')
    ap.add_argument('--no-host', action='store_true', help='skip the PCIe-inclusive measurement')
    ap.add_argument('--no-table', action='store_true', help='skip the hash-table build extra')
    ap.add_argument('--c4-clips', type=int, default=12500, help='clips per GPU of the c4_job extra (12500 = 100k over 8 GPUs)')
    ap.add_argument('--c4-batch', type=int, default=1250, help='clips per batch of the c4_job extra')
    ap.add_argument('--extras', default='ragged,c5,c4_slice,c4_job', help='which extra workloads of the N=1 line to run (comma separated)')
    ap.add_argument('--c4-ctx', type=int, default=3, help='staged contexts (batches in flight) of the c4_job extra')
    ap.add_argument('--no-overlap', action='store_true', help='one context only: batches strictly back to back')
    ap.add_argument('--inflight', type=int, default=0, help='contexts (batches in flight) when overlapping; 0 = 4 staged / 2 unstaged')
    ap.add_argument('--staged', type=int, default=-1, help='1: contexts share a spectral-stage stream and a scan-stage '
                    'stream (afp_set_stage_streams) so batch i+1\'s STFT runs beside batch i\'s scan; 0: one stream per context; '
                    '-1: staged (measured best on C3, C4 and C5)')
    ap.add_argument('--stages', type=int, default=3, help='2: spectral | scan+pair;  3: spectral | scan | pair')
    ap.add_argument('--scan-streams', type=int, default=1, help='independent scan-stage streams (contexts alternate)')
    ap.add_argument('--scan-prio', type=int, default=-1, help='torch stream priority of the scan-stage stream (-1 = high)')
    ap.add_argument('--cu-split', type=int, default=-1, help='staged mode: compute units given to the scan / pairing stages (the '
                    'spectral stage gets the rest): the stages of consecutive batches run on DISJOINT CUs instead of time-sharing '
                    'all of them; 0 = no partition; -1 = default')
    return ap.parse_args()


def start_ranks_ourselves(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves (one process per GPU, rank 0 owns stdout).
    The driver launches N > 1 through torch.distributed.run, which sets WORLD_SIZE; a plain invocation must not silently
    measure one GPU and print n_gpus: 1."""
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import socket
        import subprocess
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not os.environ.get('AFP_BENCH_ONE_GPU'):
            raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible' % (args.gpus, have))
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world_env and not (args.gpus == 1 and world_env == 1):
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world_env))


def open_bench(args):
    """This rank's process: stdout reserved for the JSON line, the process group, the GPU, NUMA binding (N > 1), the Runner,
    the clip pool."""
    B = Bench(args)
    # stdout carries exactly ONE line, the JSON of rank 0: everything else this process (or a library under it: RCCL prints
    # a version banner through C stdio at start-up) writes to file descriptor 1 goes to stderr instead
    sys.stdout.flush()
    B.json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    B.torch = torch
    B.rank = int(os.environ.get('RANK', '0'))
    B.world = int(os.environ.get('WORLD_SIZE', '1'))
    B.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # test hooks (tests/test_gpu_bench_ranks.py): AFP_BENCH_ONE_GPU=1 puts every rank on GPU 0 and AFP_BENCH_BACKEND=gloo
    # carries the collectives over gloo, so the N > 1 logic of this file runs on a one-GPU box; the driver sets neither
    if os.environ.get('AFP_BENCH_ONE_GPU'):
        B.local_rank = 0
    B.backend = os.environ.get('AFP_BENCH_BACKEND', 'nccl')
    B.dist = None
    if B.world > 1 or os.environ.get('AFP_BENCH_FORCE_DIST'):      # (the env var exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if B.backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', B.local_rank))
        else:
            dist.init_process_group(backend=B.backend)
        B.dist = dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (MI355X); there is no CPU fallback')
    if B.local_rank >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank %d has no GPU %d (%d visible)' % (B.rank, B.local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(B.local_rank)
    B.dev = torch.device('cuda', B.local_rank)
    B.rdev = B.dev if B.backend == 'nccl' else None          # where the tensors of the statistics reductions live
    # N > 1: every rank pulls ~50 GB/s of PCM out of host memory (the scaling limiter) -- keep the rank's threads
    # and, through first touch, its pinned buffers on the NUMA node its GPU hangs off.  Done before anything is allocated.
    # At N = 1 the node is reported but the process is left alone (the all-cores CPU baseline wants every core).
    B.numa_node, B.numa_cpus = numa_of_gpu(torch, B.dev)
    B.numa_bound = False
    if B.numa_cpus and (B.world > 1 or os.environ.get('AFP_BENCH_NUMA_BIND')) and not os.environ.get('AFP_BENCH_NO_NUMA_BIND'):
        try:
            os.sched_setaffinity(0, B.numa_cpus)
            B.numa_bound = True
        except OSError:
            B.numa_bound = False

    from audfprint_amd import _lib
    B.lib = _lib
    B.R = Runner(args, torch, B.dev, B.local_rank, B.dist)
    B.ex = B.R.ex
    B.BID = _lib.load().afp_build_id().decode()
    wl = dict(WORKLOADS[args.workload])
    if args.nclips:
        wl['nclips'] = args.nclips
    if args.secs:
        wl['secs'] = args.secs
    B.wl, B.nclips, B.nsamp = wl, wl['nclips'], int(round(wl['secs'] * SR))
    # synthetic input: c3 / c5 / c4_slice share the pool -- clip i of a workload with shorter clips is the first samples of pool clip i
    B.npool = min(args.pool, max(B.nclips, 1))
    B.pool = synth_pool(B.npool, B.nsamp, seed0=1000003 * B.rank)
    if not args.no_cpu:
        from oracle import afp_oracle as O          # the checker (and the CPU baseline when no reference tree is given)
        B.O = O
    return B


# ======================================================================================================================
#  Parity of a TIMED batch
# ======================================================================================================================
def guarded_pass(e, d_ptr, off, nclips, s16=False):
    """The same batch once more with the near-tie guard on (a sibling instantiation of the scan kernels, +3..5 % per step):
    it adds comparisons and changes no decision.  Used ONLY for `near_tie_units` and to show its rows equal the timed ones."""
    with guarded(e):
        e.extract_device(d_ptr, off, want_hashes=True, want_peaks=False, s16=s16)
        r = e.fetch(nclips, True, False)
        r.path = e.path_stats()
    return r


def timed_parity(B, timed, guard, ok, nchecked, how):
    """The parity object of one workload.  `timed` = the BatchResult of the LAST TIMED STEP (Runner.measure: unguarded
    kernels, the very instantiations whose time is reported); `ok` = its rows equal the oracle's on the clips checked."""
    lib = B.lib
    p = dict(clips_checked=int(nchecked), bit_exact=bool(ok), timed_variant_checked=True,
             how='rows of the LAST TIMED step (near-tie guard off: the kernel instantiations that were timed), ' + how,
             tie_prone_units=int(np.count_nonzero(timed.unit_flags & lib.UNIT_TIE)),
             timed_path=dict(compact=bool(timed.path['compact']), segments=bool(timed.path['segments']),
                             redone_dense=bool(timed.path['redone_dense'])))
    if guard is not None:
        p.update(near_tie_units=int(np.count_nonzero(guard.unit_flags & lib.UNIT_NEARTIE)), near_tie_eps=NT_EPS,
                 near_tie_redone_dense=bool(guard.path['near_tie_redone']),
                 guarded_pass_identical=same_rows(timed, guard),
                 near_tie_how='one more pass over the same batch with the guard on (afp_set_neartie_eps): counts the units in which a '
                              'decisive comparison of the threshold passes came out closer than eps; its rows equal the timed rows')
    return p


# ======================================================================================================================
#  The headline: BASELINE configs[2] (or --workload), every rank on its own clips
# ======================================================================================================================
def headline(B):
    from audfprint_amd.shard import reduce_job_stats
    import audfprint_amd
    args, wl = B.args, B.wl
    B.d_pcm, B.offsets = B.resident(B.nclips, B.nsamp)
    B.torch.cuda.synchronize()
    m = B.m = B.R.measure(wl, B.d_pcm, B.offsets, args.steps, args.warmup, overlap=not args.no_overlap, staged=args.staged,
                          inflight=args.inflight)
    elapsed, tot_hashes, audio_s_per_step = reduce_job_stats(m['elapsed'], float(m['nh']), B.nclips * wl['secs'], B.dist, B.rdev)
    ms_per_step = elapsed / args.steps * 1e3
    xrt = audio_s_per_step * args.steps / elapsed
    out = dict(metric='landmark hashes/sec (11025 Hz ingest)', value=round(tot_hashes * args.steps / elapsed, 1), unit='hashes/s',
               n_gpus=B.world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
               higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
               config=dict(workload=wl['name'], clips_per_gpu=B.nclips, clip_secs=wl['secs'], density=wl['density'],
                           fanout=wl['fanout'], shifts=wl['shifts'], sample_rate=SR, distinct_clips_per_gpu=B.npool,
                           sharding='clips/rank, no collective'),
               audio_sec_per_sec=round(xrt, 1), audio_sec_per_sec_per_gpu=round(xrt / B.world, 1),
               hashes_per_step=tot_hashes, batches_in_flight=m['nctx'], staged=m['staged'],
               cu_split=(B.R.cu_split() if m['staged'] else 0),
               ms_per_step_one_context=round(m['serial_ms'], 4), shader_mhz_under_load=m['mhz'],
               power_under_load=m['power'], build_id=B.BID,
               roofline=roofline_obj(args.workload, wl, B.nclips, B.nsamp, m['nh'], ms_per_step, m['kern_ms'], m['mhz'], B.BID),
               runtime=dict(audfprint_amd.runtime_info(), host_threads=int(B.lib.load().afp_host_threads())))
    return out


def ranks_seen(B, out):
    """which device every rank ran on: a SCALE line must show N distinct GPUs"""
    torch = B.torch
    try:
        props = torch.cuda.get_device_properties(B.dev)
        me = dict(rank=B.rank, device=B.local_rank, uuid=str(getattr(props, 'uuid', '')), name=props.name)
    except Exception as e:       # noqa: BLE001
        me = dict(rank=B.rank, device=B.local_rank, uuid='', name=repr(e))
    me.update(numa_node=B.numa_node, cpus_bound=(len(B.numa_cpus) if B.numa_bound else 0),
              cpus_allowed=len(os.sched_getaffinity(0)))
    seen = B.gather(me)
    out['ranks_seen'] = seen
    out['distinct_gpus'] = len(set((r['uuid'] or r['device']) for r in seen))


def headline_guarded(B):
    """the guarded sibling pass over the headline batch (every rank: the same calls on every rank)"""
    B.ex.set_params(**B.kw(B.wl))
    return guarded_pass(B.ex, B.d_pcm.data_ptr(), B.offsets, B.nclips)


def parity_across_ranks(B, out, guard):
    """N > 1: every rank compares 64 clips of ITS last timed step with the in-process oracle; the verdicts are AND-ed."""
    from audfprint_amd.shard import all_ranks_true
    O, timed = B.O, B.m['timed_res']
    prm = O.Params(**B.kw(B.wl))
    nchk = min(64, B.npool, B.nclips)
    ok = True
    try:
        for i in range(nchk):
            ok = ok and np.array_equal(O.extract(B.pool[i, :B.nsamp], prm)[1], timed.clip_hashes(i))
        ok = ok and (guard is None or same_rows(timed, guard))
    except Exception:       # noqa: BLE001   (a local failure is a failed check, not a missed collective)
        ok = False
    p = timed_parity(B, timed, guard, all_ranks_true(ok, B.dist, B.rdev), nchk,
                     'every rank compares %d of its own clips row by row with the in-process oracle; the verdicts are AND-ed over '
                     'the ranks (the compact path is exact by test volume, not by construction: include/afp.h)' % nchk)
    p['clips_checked_per_rank'] = p.pop('clips_checked')
    p.update(ranks=B.world, tie_prone_units_rank0=p.pop('tie_prone_units'), near_tie_units_rank0=p.pop('near_tie_units'))
    out['parity'] = p


# ======================================================================================================================
#  N > 1 extras: what the hosts sustain together; configs[3] as a job on every rank + the one exchange step
# ======================================================================================================================
def host_side_all_ranks(B, out):
    """what every rank's host side sustains at the same time (PCM staging over PCIe is the expected scaling
    limiter).  Per rank and aggregate; reported, never `value`."""
    try:
        mine = host_pipelined(B.R, B.torch, B.pool, B.npool, B.nsamp, B.wl, min(B.nclips, 256), tags=('s16',))['s16']
    except Exception as e:       # noqa: BLE001
        mine = dict(error=repr(e))
    allr = B.gather(mine)
    if B.rank == 0:
        good = [r for r in allr if 'error' not in r]
        out['host_inclusive_pipelined'] = dict(
            per_rank=allr, ranks=B.world,
            aggregate_pcie_gb_per_s=round(sum(r['pcie_gb_per_s'] for r in good), 1),
            aggregate_audio_sec_per_sec=round(sum(r['audio_sec_per_sec'] for r in good), 1),
            how='every rank at the same time: s16 PCM in pinned host memory (on the GPU\'s NUMA node when bound: ranks_seen), '
                '3 staged contexts, rows fetched to host')


def c4_job_all_ranks(B, out):
    """BASELINE configs[3] as the job it names, every rank on its own slice at the same time, then the ONE exchange step of
    the sharded job: rank 0 merges the per-rank tables in rank order (HashTable.merge, audfprint.py:226-235), tables travel
    GPU to GPU over RCCL point-to-point.  Reported, not part of `value`."""
    import threading
    from audfprint_amd.shard import merge_tables_to_rank0, reduce_job_stats, all_ranks_true
    args, rank, world, dist, R = B.args, B.rank, B.world, B.dist, B.R
    info, tb, ht, job = {}, None, None, None
    op8 = None
    try:
        ns10 = int(round(WORKLOADS['c4']['secs'] * SR))
        if B.O is not None:
            op8 = OraclePool(np.ascontiguousarray(B.pool[:, :ns10]), 8)
        np.random.seed(0)
        job, tb, ht = c4_job(R, B.torch, B.pool, B.npool, rank, args.c4_clips, args.c4_batch, args.c4_ctx, B.O, op8,
                             parity_batches=1, seed=rank)
    except Exception as e:       # noqa: BLE001
        info['error'] = 'c4_job: ' + repr(e)
        job = dict(error=repr(e))
    finally:
        if op8 is not None:
            op8.close()
    jobs = B.gather(job)
    if rank == 0:
        good = [j for j in jobs if 'error' not in j]
        agg = dict(per_rank=jobs, ranks=world)
        if good:
            tmax = max(j['job_ms'] for j in good) * 1e-3
            agg.update(aggregate_hashes_per_s=round(sum(j['hashes'] for j in good) / tmax, 1),
                       aggregate_audio_sec_per_sec=round(sum(j['clips'] for j in good) * WORKLOADS['c4']['secs'] / tmax, 1),
                       aggregate_pcie_gb_per_s=round(sum(j['pcm_bytes'] for j in good) / tmax / 1e9, 1),
                       slowest_rank_job_ms=round(tmax * 1e3, 2),
                       bit_exact=bool(all(j.get('parity', {}).get('bit_exact', False) for j in good) and len(good) == world))
        out['c4_job'] = agg
    # every rank takes part in the same collectives whatever happened locally; the exchange itself is guarded by a
    # watchdog thread: a transport that never completes must not take the throughput line down with it
    if all_ranks_true('error' not in info, dist, B.rdev):
        def _bail():
            if rank == 0:
                if getattr(B, 'line_guard', None) is not None:
                    B.line_guard.disarm()                # (this thread prints the line itself)
                out['table_merge_across_ranks'] = dict(error='exchange did not finish within 180 s; abandoned')
                B.emit(out)
            os._exit(0)
        dog = threading.Timer(180.0, _bail)
        dog.daemon = True
        dog.start()
        # rank 0 is the reference's parent, which starts empty and takes core 0's table like every other: the counts of
        # rank 0's over-full buckets are clipped to the depth on the way in (shard.merge_tables_to_rank0, fresh_parent);
        # what that removes from the grand total is known before the exchange (untimed)
        clipped = 0
        if rank == 0:
            clipped = int(np.maximum(ht.counts.astype(np.int64) - int(ht.depth), 0).sum())
        nstored = int(np.asarray(ht.hashesperid, np.int64).sum())
        # every rank's bucket counts as they stand before the exchange (untimed): rank 0 applies the reference's rule to them
        # afterwards (hash_table.py:302-321) -- the merged counts must be exactly that, bucket by bucket
        all_counts = B.gather((np.asarray(ht.counts, np.int32), int(ht.depth)))
        R.barrier()
        tm0 = time.perf_counter()
        mstats = {}
        try:
            nov = merge_tables_to_rank0(tb, dist, B.dev, stats=mstats)
        except Exception as e:
            nov = None
            info['error'] = 'merge: ' + repr(e)
        R.barrier()
        tm = time.perf_counter() - tm0
        _, tot_stored, _ = reduce_job_stats(0.0, float(nstored), 0.0, dist, B.rdev)
        dog.cancel()
        if rank == 0 and 'error' not in info:
            tb.finalize()
            tot_cnt = int(ht.counts.astype(np.int64).sum())
            # HashTable.merge on the counts alone: allvals holds min(count, depth) entries of either table (:304-305, a
            # slice stops at the row length); if they fit the count becomes their number (:315-321) -- an over-full bucket of
            # the OTHER table that meets an empty one here is clipped to the depth on the way in -- else it grows by the other
            # table's full count (:314).  The parent starts from rank 0's counts clipped the same way (fresh_parent).
            exp = np.minimum(all_counts[0][0].astype(np.int64), all_counts[0][1])
            dpt = all_counts[0][1]
            for oc_, od_ in all_counts[1:]:
                oc_ = oc_.astype(np.int64)
                n1_, n2_ = np.minimum(exp, dpt), np.minimum(oc_, od_)
                exp = np.where(oc_ == 0, exp, np.where(n1_ + n2_ > dpt, exp + oc_, n1_ + n2_))
            info = dict(ms=round(tm * 1e3, 3), ranks=world, backend=dist.get_backend(), merged_ids=len(ht.names),
                        table_total_count=tot_cnt, hashes_stored_all_ranks=int(tot_stored),
                        counts_clipped_to_depth_on_rank0=clipped,
                        counts_clipped_on_the_way_in_other_ranks=int(tot_stored) - clipped - int(exp.sum()),
                        counts_equal_reference_rule=bool(np.array_equal(exp, ht.counts.astype(np.int64))),
                        counts_add_up=bool(tot_cnt == int(exp.sum()) and len(ht.names) == world * args.c4_clips),
                        overfull_buckets_per_merge=[int(x) for x in nov],
                        transport=mstats.get('transport'), fallback=mstats.get('fallback'),
                        bytes_received_by_rank0=int(mstats.get('bytes_moved', 0)),
                        bytes_per_sending_rank=int(mstats.get('bytes_moved', 0)) // max(1, world - 1),
                        dense_table_bytes_per_rank=int((1 << 20) * 100 * 4 + (1 << 20) * 4),
                        what='counts + the filled prefixes of the rows (TableBuilder.pack), not whole tables')
    elif 'error' not in info:
        info['error'] = 'another rank failed to build its table'
    if rank == 0:
        out['table_merge_across_ranks'] = info


# ======================================================================================================================
#  N = 1: the CPU path beside the GPU, and the parity of every clip of the timed batch
# ======================================================================================================================
def cpu_baseline_one_core(B, out, guard):
    """`cpu_baseline`: the reference (AFP_REF_DIR) or its restatement over a bounded sample of the same clips, ONE thread;
    the rows it produces are compared with the LAST TIMED step's rows on the way (the compare is not part of the CPU time)."""
    args, wl, timed = B.args, B.wl, B.m['timed_res']
    kw = B.kw(wl)
    f, kind = cpu_rows_fn(B.O, kw)
    nsmp = max(1, min(args.cpu_sample, B.npool, B.nclips))
    if wl['shifts'] > 1:
        nsmp = max(1, nsmp // 8)
    cpu_hashes, ok, tc = 0, True, 0.0
    # one core, pinned (`taskset -c 0`): this thread only, for the duration of the sample
    pinned = None
    try:
        allowed = os.sched_getaffinity(0)
        with open('/proc/thread-self/stat') as fst:                   # field 39: the CPU this thread last ran on -- where the
            pinned = int(fst.read().rsplit(')', 1)[1].split()[36])    # scheduler put it, rather than a CPU 0 every container shares
        if pinned not in allowed:
            pinned = min(allowed)
        os.sched_setaffinity(0, {pinned})
    except Exception:       # noqa: BLE001
        allowed, pinned = None, None
    try:
        for i in range(nsmp):
            tc0 = time.perf_counter()
            h = f(B.pool[i, :B.nsamp])
            tc += time.perf_counter() - tc0
            cpu_hashes += len(h)
            ok = ok and np.array_equal(h, timed.clip_hashes(i))
    finally:
        if pinned is not None:
            try:
                os.sched_setaffinity(0, allowed)
            except Exception:       # noqa: BLE001
                pass
    what = ('the reference itself (dpwe/audfprint Analyzer imported unchanged from AFP_REF_DIR)' if kind == 'reference'
            else 'numpy oracle (oracle/afp_oracle.py, the restatement of the reference)')
    out['cpu_baseline'] = dict(value=round(cpu_hashes / tc, 1), unit='hashes/s', cores=1, kind=kind,
                               sample='%d of the same clips (%.0f audio-s), %s, 1 thread%s, %.1f s of extraction (parity compare '
                                      'excluded)' % (nsmp, nsmp * wl['secs'], what, '' if pinned is None else ' pinned to CPU %d' % pinned, tc),
                               audio_sec_per_sec=round(nsmp * wl['secs'] / tc, 1), host_cpus=os.cpu_count(),
                               note='kind "reference" needs the reference tree (AFP_REF_DIR=<dir>); the GPU box has none in normal '
                                    'runs.  Timed once on this class of host next to the port (one thread, 32 of these clips): '
                                    'reference 1514 x RT vs port 1465 x RT, identical rows (profiles/r03_ref_timing_on_gpu_host.log)')
    par = timed_parity(B, timed, guard, ok, nsmp, '%d clips row by row against the in-process CPU path' % nsmp)
    par['exactness'] = ('this batch ran the COMPACT path, whose filtered values differ from the reference\'s by a few ulps (the per-unit '
                        'mean is subtracted after the onset filter): identical integers are a property established by test volume '
                        '-- every clip of every bench batch, every golden, the 2048-clip near-tie sweep -- not by construction '
                        '(include/afp.h, afp_set_pipeline)')
    out['parity'] = par
    return ok


def cpu_all_cores_and_every_clip(B, out, ok_rows):
    """`cpu_baseline_allcores` over one host process per CPU the container may use (os.cpu_count() within its affinity mask and cgroup quota), one clip per task (the reference's own --ncores scheme,
    audfprint.py:249) -- and, from the same pass, the sha256 of EVERY distinct clip's rows, held against the
    digests of the last timed step."""
    args, wl, timed = B.args, B.wl, B.m['timed_res']
    nproc = max(1, args.cpu_procs or effective_cpus())
    try:
        B.opool = OraclePool(B.pool, nproc)
        nall = min(B.npool, B.nclips)
        dg, _ = B.opool.run(range(nall), B.nsamp, B.kw(wl))           # every distinct clip once: the digests of the parity check
        per = 24 if wl['shifts'] < 2 else 4
        dt, ta = B.opool.run_timed(range(nall), B.nsamp, B.kw(wl), clips_per_proc=per)
        out['cpu_baseline_allcores'] = dict(value=round(sum(d[0] for d in dt) / ta, 1), unit='hashes/s', cores=nproc,
                                            kind=B.opool.kind, host_cpus=os.cpu_count(), cpus_allowed=len(os.sched_getaffinity(0)),
                                            cgroup_cpu_limit=cgroup_cpu_limit(),
                                            audio_sec_per_sec=round(len(dt) * wl['secs'] / ta, 1),
                                            sample='%d clips per process x %d processes (os.cpu_count() = %s, affinity %d, cgroup quota %s '
                                                   'CPUs: one process per CPU this container may use) = %d clip extractions of the same pool, one '
                                                   'task per process, %.2f s' % (per, nproc, os.cpu_count(), len(os.sched_getaffinity(0)),
                                                                                 cgroup_cpu_limit(), len(dt), ta))
        gd = gpu_digests(timed, range(nall))
        bad = [i for i in range(nall) if gd[i] != dg[i]]
        # clips beyond the pool are tiled copies: their rows must equal those of their source clip
        for i in range(nall, B.nclips):
            if gpu_digests(timed, [i])[0] != gd[i % B.npool]:
                bad.append(i)
        par = out['parity']
        par.update(clips_checked=B.nclips, distinct_clips=nall, bit_exact=bool(ok_rows and not bad), mismatching_clips=bad[:8],
                   how=par['how'] + ' + all %d clips by sha256 of their rows against the CPU path run in %d host processes' % (B.nclips, nproc))
    except Exception as e:      # reported, never fatal
        out['cpu_baseline_allcores'] = dict(error=repr(e))


# ======================================================================================================================
#  N = 1 extras: the other single-GPU BASELINE configurations, same command, same contexts
# ======================================================================================================================
def extra_workload(B, key, nclips_, secs_, steps_, warmup_, nchk):
    """configs[4] (`c5`) / one GPU's slice of configs[3] resident in HBM (`c4_slice`): ms per step, roofline, parity of the
    last timed step."""
    args, R, ex = B.args, B.R, B.ex
    w = dict(WORKLOADS[key])
    ns = int(round(secs_ * SR))
    d_x, off_x = B.resident(nclips_, ns)
    mm = R.measure(w, d_x, off_x, steps_, warmup_, overlap=not args.no_overlap, keep_result=B.O is not None)
    ms = mm['elapsed'] / steps_ * 1e3
    o = dict(workload=w['name'], clips=nclips_, clip_secs=secs_, steps=steps_, warmup=warmup_, ms_per_step=round(ms, 4),
             ms_per_step_one_context=round(mm['serial_ms'], 4), batches_in_flight=mm['nctx'], staged=mm['staged'],
             hashes_per_step=int(mm['nh']), hashes_per_s=round(mm['nh'] / (ms * 1e-3), 1),
             audio_sec_per_sec=round(nclips_ * secs_ / (ms * 1e-3), 1), shader_mhz_under_load=mm['mhz'],
             power_under_load=mm['power'],
             roofline=roofline_obj(key, w, nclips_, ns, mm['nh'], ms, mm['kern_ms'], mm['mhz'], B.BID))
    if B.O is not None:
        kw = B.kw(w)
        timed = mm['timed_res']
        ex.set_params(**kw)
        guard = guarded_pass(ex, d_x.data_ptr(), off_x, nclips_)
        idx = list(range(min(nchk, B.npool, nclips_)))
        if B.opool is not None:
            dg, tx = B.opool.run(idx, ns, kw)
            ok = gpu_digests(timed, idx) == dg
            how = 'sha256 of each clip\'s rows against the CPU path run in %d host processes (%.1f s)' % (B.opool.nproc, tx)
            o['cpu_allcores_hashes_per_s'] = round(sum(d[0] for d in dg) / tx, 1)
        else:
            idx = idx[:16]
            pr = B.O.Params(**kw)
            ok = all(np.array_equal(B.O.extract(B.pool[i, :ns], pr)[1], timed.clip_hashes(i)) for i in idx)
            how = 'rows compared with the in-process oracle'
        # tiled copies beyond the pool must repeat their source clip's rows
        gd = gpu_digests(timed, range(min(B.npool, nclips_)))
        ok = ok and all(gpu_digests(timed, [i])[0] == gd[i % B.npool] for i in range(B.npool, nclips_))
        o['parity'] = timed_parity(B, timed, guard, ok, len(idx), how + '; every clip beyond the pool equals its source clip')
    del d_x
    return o


def ragged_workload(B, nclips_, steps_, warmup_, nchk, nvar=4):
    """a real file list is ragged and never repeats, so the host descriptor build (cached for
    identical batches) is part of every step.  2048 clips of 3..30 s (uniform, mean 16.5 s), `nvar` different
    length assignments resident in HBM; consecutive uses of a context see different offsets."""
    args, R, torch = B.args, B.R, B.torch
    w = dict(WORKLOADS['c3'])
    rng = np.random.RandomState(1000003 * B.rank + 7)
    variants = []
    d_pool = torch.from_numpy(B.pool).to(B.dev)
    for v in range(nvar):
        lens = rng.randint(3 * SR, 30 * SR + 1, size=nclips_).astype(np.int64)
        src = (np.arange(nclips_) + 17 * v) % B.npool
        off = np.zeros(nclips_ + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        d = torch.empty(int(off[-1]), dtype=torch.float32, device=B.dev)
        for i in range(nclips_):
            d[off[i]:off[i + 1]] = d_pool[src[i], :lens[i]]
        variants.append((d, off, lens, src))
    del d_pool
    exs = R.contexts(4 if not args.no_overlap else 1, 1)
    for e in exs:
        e.set_params(**B.kw(w))
    variant_of = lambda k: (k // len(exs) + k) % nvar

    def run_steps(n):
        fl, nh_, audio = [], 0, 0.0
        for k in range(n):
            e = exs[k % len(exs)]
            if len(fl) == len(exs):
                nh_ += fl.pop(0).counts()[0]
            d, off, lens, _ = variants[variant_of(k)]
            e.extract_device(d.data_ptr(), off, want_hashes=True, want_peaks=False)
            audio += float(lens.sum()) / SR
            fl.append(e)
        for e in fl:
            nh_ += e.counts()[0]
        return nh_, audio
    run_steps(max(warmup_, nvar * len(exs)))          # every context has sized its workspace for every variant
    R.barrier()
    t0_ = time.perf_counter()
    nh_, audio = run_steps(steps_)
    R.barrier()
    el = time.perf_counter() - t0_
    last, vlast = exs[(steps_ - 1) % len(exs)], variant_of(steps_ - 1)
    timed = last.fetch(nclips_, True, False)           # the rows of the last timed step, before anything else runs there
    timed.path = last.path_stats()
    o = dict(workload='%d clips of 3..30 s (uniform), density 20, fanout 3; %d different length assignments, no two '
                      'consecutive batches of a context alike (descriptor build in every step)' % (nclips_, nvar),
             clips=nclips_, steps=steps_, ms_per_step=round(el / steps_ * 1e3, 4), batches_in_flight=len(exs),
             hashes_per_s=round(nh_ / el, 1), audio_sec_per_sec=round(audio / el, 1),
             audio_sec_per_step=round(audio / steps_, 1))
    if B.O is not None and B.opool is not None:
        d, off, lens, src = variants[vlast]
        B.ex.set_params(**B.kw(w))
        guard = guarded_pass(B.ex, d.data_ptr(), off, nclips_)
        idx = list(range(min(nchk, nclips_)))
        dg, tx = B.opool.run_var([src[i] for i in idx], [lens[i] for i in idx], B.kw(w))
        o['parity'] = timed_parity(B, timed, guard, gpu_digests(timed, idx) == dg, len(idx),
                                   'length assignment %d: sha256 of each clip\'s rows against the CPU path run in %d host processes'
                                   % (vlast, B.opool.nproc))
    return o


def host_inclusive(B, out):
    """PCIe-inclusive rate (host buffers in, host arrays out): reported, never `value`."""
    ex, wl, nclips, nsamp = B.ex, B.wl, B.nclips, B.nsamp
    ex.set_params(**B.kw(wl))
    nh_clips = min(nclips, 256)
    h_pcm = np.ascontiguousarray(B.pool[np.arange(nh_clips) % B.npool, :nsamp].reshape(-1))
    h_off = np.arange(nh_clips + 1, dtype=np.int64) * nsamp
    h16 = np.round(h_pcm * 32768).astype(np.int16)
    inc = {}
    for tag, arr in (('float32', h_pcm), ('s16', h16)):
        ex.extract(pcm=arr, offsets=h_off)
        th0 = time.perf_counter()
        for _ in range(3):
            rr = ex.extract(pcm=arr, offsets=h_off)
        th = (time.perf_counter() - th0) / 3
        inc[tag] = dict(ms_per_batch=round(th * 1e3, 3), clips=nh_clips, hashes_per_s=round(len(rr.hashes) / th, 1),
                        audio_sec_per_sec=round(nh_clips * wl['secs'] / th, 1))
    out['host_inclusive'] = inc
    # the same with the upload of batch i + 1 under the kernels of batch i (pinned buffers, three staged contexts)
    try:
        out['host_inclusive_pipelined'] = host_pipelined(B.R, B.torch, B.pool, B.npool, nsamp, wl, nh_clips)
    except Exception as e:       # noqa: BLE001
        out['host_inclusive_pipelined'] = dict(error=repr(e))


def analyzer_path(B):
    """The drop-in's own call pattern: ONE file per Analyzer call (audfprint.py:164-165, 177-182), decode excluded.  The
    rows compared with the oracle are those of the last timed call."""
    from audfprint_amd import audfprint_analyze as AA
    from oracle import afp_oracle as O

    class _MemAnalyzer(AA.Analyzer):
        """wavfile2hashes with the decode replaced by a waveform already in memory (the decoder is the user's audio_read
        module: ffmpeg, not part of the path)."""
        clip = None

        def _read_audio(self, filename):
            return self.clip, SR

    ap_ = {}
    for secs_, ncall in ((10.0, 100), (300.0, 20)):
        an = _MemAnalyzer()
        an.clip = O.synth_noise(77, secs_)
        for _ in range(3):
            hh = an.wavfile2hashes('mem.wav')
        ta0 = time.perf_counter()
        for _ in range(ncall):
            hh = an.wavfile2hashes('mem.wav')
        ta = (time.perf_counter() - ta0) / ncall
        ref = O.extract(an.clip, O.Params())[1]
        sg = B.ex.seg_stats()                      # (the Analyzer runs on this process's Extractor: the last call's cut)
        ap_['%ds' % int(secs_)] = dict(ms_per_call=round(ta * 1e3, 4), calls=ncall, hashes=int(len(hh)),
                                       audio_sec_per_sec=round(secs_ / ta, 1), bit_exact=bool(np.array_equal(hh, ref)),
                                       segments=sg['segments'], segments_rerun=sg['rerun_fwd'] + sg['rerun_bwd'],
                                       cut=dict(own_frames=sg['seg_len'], warm_up_frames=sg['seg_warm']),
                                       short_cut_backoffs=sg['short_cut_backoffs'])
    # the bulk form of the same method: 256 ten-second files in one call (one launch of the hot path; dense / compact path by size)
    class _MemAnalyzerMany(AA.Analyzer):
        clips = {}

        def _read_audio(self, filename):
            return self.clips[filename], SR

    anm = _MemAnalyzerMany()
    names = ['mem%03d.wav' % i for i in range(256)]
    anm.clips = dict((n, O.synth_noise(9000 + i, 10.0)) for i, n in enumerate(names))
    for _ in range(2):
        many = anm.wavfiles2hashes(names)
    tb0 = time.perf_counter()
    for _ in range(5):
        many = anm.wavfiles2hashes(names)
    tbk = (time.perf_counter() - tb0) / 5
    prm0 = O.Params()
    chk = [0, 17, 101, 255]
    ap_['bulk_256x10s'] = dict(ms_per_call=round(tbk * 1e3, 3), ms_per_file=round(tbk * 1e3 / len(names), 4), files=len(names),
                               audio_sec_per_sec=round(len(names) * 10.0 / tbk, 1),
                               bit_exact=bool(all(np.array_equal(many[i], O.extract(anm.clips[names[i]], prm0)[1]) for i in chk)),
                               files_checked=len(chk),
                               how='Analyzer.wavfiles2hashes(256 files): the same list a loop over wavfile2hashes returns, one launch')
    ap_['how'] = ('Analyzer.wavfile2hashes per file (decode excluded): host PCM in, (N,2) int32 rows out, one call at a '
                  'time, through the segment-parallel scan; files of up to 1000 frames take the short cut (32 + 96 frames per '
                  'segment) while it converges (afp_get_seg_stats); bit_exact = the rows of the LAST timed call against the oracle')
    return ap_


def table_build(B):
    """hash-table build (store + merge) of the headline batch (reported as an extra)."""
    import random
    from audfprint_amd.table import TableBuilder
    from oracle import afp_oracle as O
    args, ex, wl, nclips, torch = B.args, B.ex, B.wl, B.nclips, B.torch
    ex.set_params(**B.kw(wl))
    ex.extract_device(B.d_pcm.data_ptr(), B.offsets, want_hashes=True, want_peaks=False)
    res_t = ex.fetch(nclips, True, False)
    ht = _TableArrays(hashbits=20, depth=100)                # the reference HashTable's fields, nothing else
    tb = TableBuilder(ht, ex)
    tnames = ['clip%06d' % i for i in range(nclips)]
    random.seed(0)
    torch.cuda.synchronize()
    tt0 = time.perf_counter()
    novf = tb.store_batch(tnames, offsets=res_t.hash_offsets)   # rows stay in HBM
    tt1 = time.perf_counter()
    # HashTable.merge (hash_table.py:291-323) of a second per-GPU table (the same batch under other names:
    # what the parent of `new --ncores N` does with its workers' tables, audfprint.py:226-235)
    ex2 = B.R.contexts(2, 0)[1]
    ex2.set_params(**B.kw(wl))
    ex2.extract_device(B.d_pcm.data_ptr(), B.offsets, want_hashes=True, want_peaks=False)
    res_2 = ex2.fetch(nclips, True, False)
    ht2 = _TableArrays(hashbits=20, depth=100)
    tb2 = TableBuilder(ht2, ex2)
    random.seed(0)                                        # (the same draws as the first table: its oracle twin is then a copy)
    tb2.store_batch(['other%06d' % i for i in range(nclips)], offsets=res_2.hash_offsets)
    np.random.seed(0)
    torch.cuda.synchronize()
    tm0 = time.perf_counter()
    nmov = tb.merge(ht2, other_device_ptrs=tb2.device_ptrs())
    tm1 = time.perf_counter()
    tb.finalize()
    tt2 = time.perf_counter()
    o = dict(hashes=int(len(res_t.hashes)), store_ms=round((tt1 - tt0) * 1e3, 3),
             store_kernels_ms=round(tb.seconds['store'] * 1e3, 3), overflow_replay_ms=round(tb.seconds['replay'] * 1e3, 3),
             merge_ms=round((tm1 - tm0) * 1e3, 3), merge_overfull_buckets=int(nmov),
             download_ms=round((tt2 - tm1) * 1e3, 3), overflow_events=int(novf),
             gpu_hashes_per_s=round(len(res_t.hashes) / (tt1 - tt0), 1),
             merged_ids=len(ht.names), table_total_count=int(ht.counts.sum()),
             table_nonzero_buckets=int(np.count_nonzero(ht.counts)))
    if not args.no_cpu:
        # the reference's per-hash Python loop (the oracle's restatement of HashTable.store / merge) over the SAME rows,
        # names, seeds: timed, and the tables compared
        import copy
        ref_t = O.OracleHashTable(hashbits=20, depth=100)
        rr = random.Random(0)
        tc0 = time.perf_counter()
        for i in range(nclips):
            ref_t.store(tnames[i], res_t.clip_hashes(i), rr)
        tc = time.perf_counter() - tc0
        ref_2 = copy.deepcopy(ref_t)
        ref_2.names = ['other%06d' % i for i in range(nclips)]
        tc1 = time.perf_counter()
        ref_t.merge(ref_2, np.random.RandomState(0))
        tcm = time.perf_counter() - tc1
        ok = (np.array_equal(ht.table, ref_t.table) and np.array_equal(ht.counts, ref_t.counts) and ht.names == ref_t.names and
              np.array_equal(np.asarray(ht.hashesperid, np.int64), np.asarray(ref_t.hashesperid, np.int64)))
        o.update(cpu_loop_hashes_per_s=round(len(res_t.hashes) / tc, 1), cpu_store_s=round(tc, 2), cpu_merge_s=round(tcm, 2),
                 parity=dict(bit_exact=bool(ok), clips=2 * nclips, rows=2 * int(len(res_t.hashes)),
                             how='store of this batch (random.seed(0)) + merge of a second table built from the same rows '
                                 '(np.random.seed(0)): table, counts, names, hashesperid equal OracleHashTable.store / .merge'))
    return o


class _MatcherSwitches(object):
    """The attributes of audfprint_match.Matcher that match_hashes reads (audfprint_match.py:93-123), at their defaults."""
    window, threshcount, search_depth, max_alignments_per_id = 1, 5, 100, 100
    exact_count, find_time_range, time_quantile = False, False, 0.02


def match_queries(B, nref=512, nq=64, qsecs=8.0):
    """SURVEY §8f row f4 measured: HashTable.get_hits (hash_table.py:150-176) + the matcher's vote counting
    (Matcher.match_hashes, audfprint_match.py:314-352) over a device-resident table.  Reference set: the first `nref` pool
    clips (30 s, density 20) extracted and stored on the GPU; queries: `nq` excerpts of `qsecs` seconds cut from reference
    clips at an offset that is not a multiple of the hop, with white noise added at -6 dB of the clip's level, fingerprinted the
    way `audfprint match` does it (4 shifts, audfprint.py:295-297).  Timed: match_hashes per query on the GPU table; the same
    queries through the oracle's restatement of get_hits + the matcher on the host copy of the SAME table; every result array
    compared."""
    import random
    from audfprint_amd.table import TableBuilder
    from audfprint_amd import match as MM
    O, ex, torch = B.O, B.ex, B.torch
    nref = min(nref, B.npool, B.nclips)
    # ---- the reference table
    ex.set_params(density=20.0, maxpairsperpeak=3, shifts=1)
    d_ref, off_ref = B.resident(nref, B.nsamp)
    torch.cuda.synchronize()
    ex.extract_device(d_ref.data_ptr(), off_ref, want_hashes=True, want_peaks=False)
    r = ex.fetch(nref, True, False)
    ht = _TableArrays(hashbits=20, depth=100)
    tb = TableBuilder(ht, ex)
    random.seed(0)
    tb.store_batch(['ref%05d' % i for i in range(nref)], offsets=r.hash_offsets)
    # ---- the queries (host side: what a user's files would be), 4 shifts like `audfprint match`
    rng = np.random.RandomState(4242 + B.rank)
    qn = int(round(qsecs * SR))
    src = rng.randint(0, nref, nq)
    qoff = rng.randint(SR, B.nsamp - qn - SR, nq) | 1                # odd offsets: never frame-aligned with the reference
    clips = []
    for i in range(nq):
        c = B.pool[src[i], qoff[i]:qoff[i] + qn].astype(np.float64)
        c = c + rng.randn(qn) * (0.5 * c.std())
        clips.append((np.round(np.clip(c, -1, 1) * 32767) / 32768.0).astype(np.float32))
    ex2 = B.R.contexts(2, 0)[1]                                       # (the table lives on ex: queries are fingerprinted on another context)
    ex2.set_params(density=20.0, maxpairsperpeak=3, shifts=4)
    rq = ex2.extract(clips=clips, want_hashes=True, want_peaks=False)
    queries = [rq.clip_hashes(i).copy() for i in range(nq)]
    m = _MatcherSwitches()
    for q in queries[:4]:
        MM.match_hashes(m, tb, q)                                     # warm-up (buffers of the hit / vote kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = [MM.match_hashes(m, tb, q) for q in queries]
    tg = (time.perf_counter() - t0) / nq
    top1 = sum(1 for i in range(nq) if len(got[i]) and int(got[i][0, 0]) == int(src[i]))
    # the alignment the matcher reports for the best match: skew = reference frame - query frame = round(offset / hop)
    skew_ok = sum(1 for i in range(nq) if len(got[i]) and int(got[i][0, 0]) == int(src[i]) and abs(int(got[i][0, 2]) - qoff[i] / 256.0) <= 1.0)
    o = dict(workload='%d queries of %.0f s (excerpts of the reference clips at odd sample offsets + white noise at -6 dB, 4 shifts) against '
                      'a device-resident table of %d x %.0f s clips (density 20): HashTable.get_hits + Matcher.match_hashes, default switches'
                      % (nq, qsecs, nref, B.wl['secs']),
             queries=nq, reference_clips=nref, table_entries=int(r.hash_offsets[-1]),
             query_hashes_mean=round(float(np.mean([len(q) for q in queries])), 1),
             ms_per_query=round(tg * 1e3, 4), queries_per_s=round(1.0 / tg, 1),
             top1_is_the_source_clip=top1, top1_alignment_within_one_frame=skew_ok)
    if O is not None:
        tb.finalize()
        ref = O.OracleHashTable(hashbits=20, depth=100)
        ref.table, ref.counts, ref.names = ht.table, ht.counts, list(ht.names)
        ref.hashesperid = np.asarray(ht.hashesperid)
        nchk = min(nq, 16)
        tc0 = time.perf_counter()
        want = [O.match_hashes(ref, q, m.window, m.threshcount, m.search_depth, m.max_alignments_per_id) for q in queries[:nchk]]
        tc = (time.perf_counter() - tc0) / nchk
        ok = all(w.shape == g.shape and np.array_equal(np.asarray(w, np.int64), np.asarray(g, np.int64)) for w, g in zip(want, got[:nchk]))
        o.update(cpu_ms_per_query=round(tc * 1e3, 2),
                 parity=dict(queries_checked=nchk, bit_exact=bool(ok),
                             how='result arrays [id, count, skew, raw count, rank, 0, 0] of the first %d queries equal the oracle\'s '
                                 'get_hits + match_hashes (hash_table.py:150-176, audfprint_match.py:124-147, 241-352 restated) over the '
                                 'host copy of the same table' % nchk))
    del d_ref
    return o


def c2_single_clip(B):
    """configs[1]: one 300 s clip resident in HBM (latency-bound; reported, not the headline); the KAT of SURVEY §8c."""
    torch, ex = B.torch, B.ex
    c2 = synth_pool(1, 300 * SR, seed0=0)
    w2 = WORKLOADS['c2']
    d_c2 = torch.from_numpy(c2).to(B.dev).view(-1)
    off2 = np.array([0, 300 * SR], dtype=np.int64)
    m2 = B.R.measure(w2, d_c2, off2, 10, 3, overlap=False)
    t2 = m2['serial_ms'] * 1e-3
    n2 = m2['nh']
    o2 = dict(workload=w2['name'], ms=round(t2 * 1e3, 3), hashes=int(n2),
              hashes_per_s=round(n2 / t2, 1), audio_sec_per_sec=round(300.0 / t2, 1),
              kat_hashes_expected=19571,
              roofline=roofline_obj('c2', w2, 1, 300 * SR, n2, t2 * 1e3, m2['kern_ms'], None, B.BID))
    if B.O is not None:
        timed = m2['timed_res']
        tq = time.perf_counter()
        h2 = B.O.extract(c2[0], B.O.Params(density=20.0, maxpairsperpeak=3, shifts=1))[1]
        tq = time.perf_counter() - tq
        ex.set_params(density=20.0, maxpairsperpeak=3, shifts=1)
        guard = guarded_pass(ex, d_c2.data_ptr(), off2, 1)
        sg = ex.seg_stats()       # segment-parallel scan of the long unit: segments, re-runs, final check
        p = timed_parity(B, timed, guard, np.array_equal(h2, timed.clip_hashes(0)), 1, 'row by row against the in-process oracle')
        p.update(sha16=_digest(timed.clip_hashes(0)), kat_sha16_expected='04f537147efd7b79', cpu_oracle_s=round(tq, 3))
        o2['parity'] = p
        o2['segments'] = sg['segments']
        o2['segments_rerun'] = dict(forward=sg['rerun_fwd'], backward=sg['rerun_bwd'])
        o2['segment_check_failed'] = sg['failed']
    return o2


def single_gpu_extras(B, out, guard):
    """Everything the N = 1 line carries besides the headline (rank 0 only).  No extra may cost the line: each runs behind
    try / except and leaves `<name>_error` instead of its object if it fails."""
    args = B.args

    def attempt(key, fn):
        try:
            r = fn()
            if r is not None:
                out[key] = r
        except Exception as e:       # noqa: BLE001   (reported; the contract fields are already in `out`)
            out[key + '_error'] = repr(e)

    if B.O is not None:
        ok_rows = [False]

        def one_core():
            ok_rows[0] = cpu_baseline_one_core(B, out, guard)
        attempt('cpu_baseline', one_core)
        if not args.no_cpu_all and 'parity' in out:
            attempt('cpu_baseline_allcores', lambda: cpu_all_cores_and_every_clip(B, out, ok_rows[0]))
    if not args.no_extras and args.workload == 'c3' and not args.nclips and not args.secs:
        want_x = set(x.strip() for x in args.extras.split(','))
        if 'ragged' in want_x:
            attempt('ragged', lambda: ragged_workload(B, 2048, 20, 4, 128))
        if 'c5' in want_x:
            attempt('c5', lambda: extra_workload(B, 'c5', 1024, 30.0, 20, 4, 64))
        if 'c4_slice' in want_x:
            attempt('c4_slice', lambda: extra_workload(B, 'c4', 12500, 10.0, 20, 4, 256))
        if not args.no_table and 'c4_job' in want_x:
            attempt('c4_job', lambda: c4_job(B.R, B.torch, B.pool, B.npool, B.rank, args.c4_clips, args.c4_batch, args.c4_ctx, B.O, B.opool,
                                             whole_job_parity=B.O is not None)[0])
    if B.opool is not None:
        try:
            B.opool.close()
        except Exception:       # noqa: BLE001
            pass
        B.opool = None
    if not args.no_host:
        attempt('host_inclusive', lambda: host_inclusive(B, out))
        attempt('analyzer_path', lambda: analyzer_path(B))
    if not args.no_table:
        attempt('table_build', lambda: table_build(B))
        attempt('match_queries', lambda: match_queries(B))
    if not args.no_c2 and args.workload != 'c2':
        attempt('c2_single_clip', lambda: c2_single_clip(B))


class LineGuard(object):
    """N > 1, rank 0: the contract line must survive the extras.  host_side_all_ranks / c4_job_all_ranks exercise what no box of
    this pool could run -- RCCL point-to-point between GPUs, out of library memory -- behind try / except and a watchdog for
    hangs; a hard crash (a fault inside the runtime or RCCL) would still take the process down before the line is printed.
    So before the extras a child is forked that holds the line AS IT STANDS (headline, ranks_seen, parity: everything the
    contract asks for) and does nothing but wait on a pipe: the word `done` lets it go silently; the pipe closing without it
    -- this process died -- makes it print that line, marked `extras_crashed`.  The child makes no HIP / torch call, only
    read / write / _exit."""

    def __init__(self, B, out):
        payload = (json.dumps(dict(out, extras_crashed='rank 0 died inside the N > 1 extras (host-inclusive leg / c4_job / '
                                                        'cross-rank table merge); this is the line as it stood before them')) + '\n').encode()
        sys.stdout.flush()
        sys.stderr.flush()
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            try:
                os.close(w)
                word = os.read(r, 8)
                if word != b'done':
                    os.write(B.json_fd, payload)
            finally:
                os._exit(0)
        os.close(r)
        self.w, self.pid = w, pid

    def disarm(self):
        if self.w is not None:
            try:
                os.write(self.w, b'done')
                os.close(self.w)
                os.waitpid(self.pid, 0)
            except OSError:
                pass
            self.w = None


def main():
    args = parse_args()
    start_ranks_ourselves(args)
    B = open_bench(args)
    out = headline(B)                       # W warm-up + K timed steps of the hot path; the contract fields of the line
    ranks_seen(B, out)
    guard = None
    if B.O is not None:
        try:
            guard = headline_guarded(B)
        except Exception as e:       # noqa: BLE001   (no collective inside: a rank that fails here still meets the others below)
            out['guarded_pass_error'] = repr(e)
    if B.world > 1:
        if B.O is not None:
            parity_across_ranks(B, out, guard)
        B.line_guard = LineGuard(B, out) if (B.rank == 0 and not os.environ.get('AFP_BENCH_NO_LINE_GUARD')) else None
        if os.environ.get('AFP_BENCH_CRASH_IN_EXTRAS') and B.rank == 0:      # (test hook: tests/test_gpu_bench_ranks.py)
            import signal
            os.kill(os.getpid(), signal.SIGSEGV)
        if not args.no_host:
            host_side_all_ranks(B, out)
        if not args.no_table:
            c4_job_all_ranks(B, out)
        if B.line_guard is not None:
            B.line_guard.disarm()
    elif B.rank == 0:
        single_gpu_extras(B, out, guard)
    if B.rank == 0:
        B.emit(out)
    os.close(B.json_fd)
    if B.dist is not None:
        B.dist.barrier()
        B.dist.destroy_process_group()


if __name__ == '__main__':
    main()
