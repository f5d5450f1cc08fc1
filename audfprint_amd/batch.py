"""Batch API over libafp_hip.so: many clips per launch (what the bench and bulk ingest use).

The reference has no batch entry point -- its CLI feeds one file at a time
(audfprint.py:164-165,177-182).  `Extractor` is the one place that computes the host-side
constants (bit-identical to the reference's numpy values) and owns the per-process,
per-GPU library handle.  It holds no state that needs pickling: the handle is created
lazily in whichever process first uses it (fork / joblib safe)."""
import ctypes as C
import os

import numpy as np

from . import _lib

OVERSAMP = 1            # audfprint_analyze.py:59
HPF_POLE = 0.98         # audfprint_analyze.py:67
N_FFT = 512             # audfprint_analyze.py:64
N_HOP = 256             # audfprint_analyze.py:65


class BatchResult(object):
    """CSR results of one batch.
    hashes            (N,2) int32 rows (time, hash), sorted unique per clip
    hash_offsets      (nclips+1,) int64 row offsets per clip
    peaks             (P,2) int32 rows (col, bin) per unit (unit = clip*shifts + shift)
    peak_offsets      (nunits+1,) int64
    unit_flags        (nunits,) int32, _lib.UNIT_* bits"""

    def __init__(self):
        self.hashes = None
        self.hash_offsets = None
        self.peaks = None
        self.peak_offsets = None
        self.unit_flags = None
        self.nclips = 0
        self.shifts = 1

    def clip_hashes(self, i):
        return self.hashes[self.hash_offsets[i]:self.hash_offsets[i + 1]]

    def unit_peaks(self, clip, shift=0):
        u = clip * self.shifts + shift
        return self.peaks[self.peak_offsets[u]:self.peak_offsets[u + 1]]


def host_constants(density, n_fft, n_hop, f_sd, shifts):
    """The float constants of the path, computed with the reference's own numpy expressions."""
    a_dec = (1 - 0.01 * (density * np.sqrt(n_hop / 352.8) / 35)) ** (1 / OVERSAMP)   # audfprint_analyze.py:277
    window = np.ascontiguousarray(np.hanning(n_fft + 2)[1:-1], dtype=np.float64)      # :279
    npoints = 256
    sp_vals = np.exp(-0.5 * ((np.arange(-npoints, npoints + 1) / f_sd) ** 2))        # :191-192
    gauss = np.ascontiguousarray(sp_vals[npoints:2 * npoints], dtype=np.float64)
    nsh = 1 if (shifts is None or shifts < 2) else int(shifts)                       # :369
    offs = [0] if nsh == 1 else [int(s / shifts * n_hop) for s in range(nsh)]        # :375
    return float(a_dec), window, gauss, offs


def cu_range_stream(device, first_cu, n_cus):
    """A raw hipStream_t (int) whose kernels run only on CUs [first_cu, first_cu + n_cus) -- for
    Extractor.set_stage_streams (afp_stream_create_cu_range).  Release with destroy_stream()."""
    s = C.c_void_p()
    _lib.check(_lib.load().afp_stream_create_cu_range(int(device), int(first_cu), int(n_cus), C.byref(s)),
               'afp_stream_create_cu_range')
    return s.value


class _Pinned(object):
    """Owner of one afp_pinned_alloc block; the numpy arrays made over it keep it alive (ndarray.base -> ctypes buffer -> this)."""

    def __init__(self, device, nbytes):
        self.lib = _lib.load()
        p = C.c_void_p()
        _lib.check(self.lib.afp_pinned_alloc(int(device), int(nbytes), C.byref(p)), 'afp_pinned_alloc')
        self.ptr, self.nbytes, self.pid = p.value, int(nbytes), os.getpid()

    def __del__(self):
        try:
            if self.ptr and self.pid == os.getpid():
                self.lib.afp_pinned_free(C.c_void_p(self.ptr))
            self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32, device=0):
    """numpy array in page-locked host memory (afp_pinned_alloc): what Extractor.submit uploads asynchronously.  For hosts
    without torch (torch.empty(...).pin_memory().numpy() is the same kind of memory)."""
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if shape else 1
    own = _Pinned(device, max(1, n * dt.itemsize))
    buf = (C.c_char * own.nbytes).from_address(own.ptr)
    buf._afp_owner = own                   # every array / view made from `buf` keeps the block alive; the last one frees it
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


def destroy_stream(stream):
    _lib.check(_lib.load().afp_stream_destroy(C.c_void_p(stream)), 'afp_stream_destroy')


class Extractor(object):
    _instances = {}
    _inherited = []

    @classmethod
    def get(cls, device=0):
        """Per-process, per-device singleton (created after fork, never pickled)."""
        key = (os.getpid(), int(device))
        inst = cls._instances.get(key)
        if inst is None:
            inst = cls(device)
            # entries inherited from a parent process across fork() stay referenced forever (cls._inherited):
            # their HIP handles belong to the parent's context and must never be destroyed from this process
            cls._inherited.extend(v for k, v in cls._instances.items() if k[0] != os.getpid())
            cls._instances = {k: v for k, v in cls._instances.items() if k[0] == os.getpid()}
            cls._instances[key] = inst
        return inst

    def __init__(self, device=0):
        _lib.configure_runtime()               # (first context of the process: before this library's first HIP call)
        self.lib = _lib.load()
        _lib._runtime['touched'] = True
        if self.lib.afp_device_count() <= 0:
            raise _lib.AfpError('audfprint_amd: no HIP device visible -- the extraction path needs an '
                                'MI355X (gfx950); there is no CPU fallback')
        h = C.c_void_p()
        _lib.check(self.lib.afp_create(int(device), C.byref(h)), 'afp_create')
        self.h = h
        self.pid = os.getpid()                  # a HIP context does not survive fork(): only this process may use / free h
        self.device = int(device)
        self._pkey = None
        self.shifts = 1
        self.K = 5
        self.last_nclips = 0                    # clips of the batch last queued (TableBuilder.store_batch checks names against it)

    def close(self):
        if getattr(self, 'h', None):
            if getattr(self, 'pid', None) == os.getpid():
                self.lib.afp_destroy(self.h)
            self.h = None                       # (a forked child just forgets the parent's handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -----------------------------------------------------------------------
    def set_params(self, density=20.0, maxpksperframe=5, maxpairsperpeak=3, f_sd=30.0, shifts=1,
                   targetdf=31, mindt=2, targetdt=63, n_fft=N_FFT, n_hop=N_HOP):
        if int(n_fft) != N_FFT or int(n_hop) != N_HOP:
            raise ValueError('audfprint_amd supports n_fft=512, n_hop=256 only (audfprint.py:292-293 hard-wires them)')
        key = (float(density), int(maxpksperframe), int(maxpairsperpeak), float(f_sd),
               None if shifts is None else int(shifts), int(targetdf), int(mindt), int(targetdt))
        if key == self._pkey:
            return
        a_dec, window, gauss, offs = host_constants(float(density), N_FFT, N_HOP, float(f_sd), shifts)
        if len(offs) > _lib.AFP_MAX_SHIFTS:
            raise ValueError('shifts > %d not supported' % _lib.AFP_MAX_SHIFTS)
        if not (1 <= int(maxpksperframe) <= _lib.AFP_MAX_PKS):
            raise ValueError('maxpksperframe must be in 1..%d' % _lib.AFP_MAX_PKS)
        p = _lib.AfpParams()
        p.a_dec = a_dec
        p.hpf_pole = HPF_POLE ** (1 / OVERSAMP)                                       # :294
        p.maxpksperframe = int(maxpksperframe)
        p.maxpairsperpeak = int(maxpairsperpeak)
        p.targetdf, p.mindt, p.targetdt = int(targetdf), int(mindt), int(targetdt)
        p.nshifts = len(offs)
        for i, o in enumerate(offs):
            p.shift_offsets[i] = o
        p.window = window.ctypes.data_as(C.POINTER(C.c_double))
        p.gauss = gauss.ctypes.data_as(C.POINTER(C.c_double))
        _lib.check(self.lib.afp_set_params(self.h, C.byref(p)), 'afp_set_params')
        self._pkey = key
        self.shifts = len(offs)
        self.K = int(maxpksperframe)

    def set_params_from(self, an):
        """Read the Analyzer-style attributes at call time (they are mutated after __init__,
        audfprint.py:285-298)."""
        self.set_params(density=an.density, maxpksperframe=an.maxpksperframe,
                        maxpairsperpeak=an.maxpairsperpeak, f_sd=an.f_sd, shifts=an.shifts,
                        targetdf=an.targetdf, mindt=an.mindt, targetdt=an.targetdt,
                        n_fft=an.n_fft, n_hop=an.n_hop)

    # ---- extraction -------------------------------------------------------------------------
    @staticmethod
    def pack(clips, dtype=np.float32):
        """list of 1-D arrays -> (pcm of `dtype`, int64 offsets)."""
        lens = np.array([len(c) for c in clips], dtype=np.int64)
        offsets = np.zeros(len(clips) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        pcm = np.empty(int(offsets[-1]), dtype=dtype)
        for c, o in zip(clips, offsets[:-1]):
            pcm[o:o + len(c)] = np.asarray(c, dtype=dtype)
        return pcm, offsets

    def _flags(self, want_hashes, want_peaks, debug):
        return ((_lib.WANT_HASHES if want_hashes else 0) | (_lib.WANT_PEAKS if want_peaks else 0)
                | (_lib.KEEP_DEBUG if debug else 0))

    def extract(self, clips=None, pcm=None, offsets=None, want_hashes=True, want_peaks=False, debug=False):
        """Run the hot path over host-resident clips; returns a BatchResult of numpy arrays."""
        if clips is not None and len(clips) == 1:
            # one file per call (the Analyzer class): the clip IS the batch -- no copy into a packed buffer (0.2 ms of a
            # 300 s file's 1.45 ms, tools/analyzer_breakdown.py), offsets valid by construction
            a = np.asarray(clips[0])
            dt = a.dtype if a.dtype in (np.int16, np.float32) else np.float64
            pcm = np.ascontiguousarray(a.reshape(-1), dtype=dt)
            offsets = np.array([0, pcm.size], dtype=np.int64)
            nclips = 1
        else:
            if clips is not None:
                kinds = set(np.asarray(c).dtype for c in clips)
                # all int16 -> raw s16 path; all float32 -> float32; anything else -> float64 (exact for both)
                dt = np.int16 if kinds == {np.dtype(np.int16)} else np.float32 if kinds <= {np.dtype(np.float32)} else np.float64
                pcm, offsets = self.pack(clips, dt)
            pcm = np.asarray(pcm).reshape(-1)
            offsets, nclips = self._check_offsets(offsets, pcm.size)
        flags = self._flags(want_hashes, want_peaks, debug)
        self.last_nclips = nclips
        if pcm.dtype == np.int16:
            # raw s16le samples: converted on the GPU exactly like audio_read.buf_to_float (audio_read.py:121-145)
            pcm = np.ascontiguousarray(pcm)
            _lib.check(self.lib.afp_extract_host_s16(self.h, pcm.ctypes.data, offsets.ctypes.data, nclips, flags), 'afp_extract_host_s16')
        elif pcm.dtype == np.float64:
            # a float64 waveform stays float64 (the reference's find_peaks never rounds it: stft.py:87-93)
            pcm = np.ascontiguousarray(pcm)
            _lib.check(self.lib.afp_extract_host_f64(self.h, pcm.ctypes.data, offsets.ctypes.data, nclips, flags), 'afp_extract_host_f64')
        else:
            pcm = np.ascontiguousarray(pcm, dtype=np.float32)
            _lib.check(self.lib.afp_extract_host(self.h, pcm.ctypes.data, offsets.ctypes.data, nclips, flags), 'afp_extract_host')
        return self.fetch(nclips, want_hashes, want_peaks)

    @staticmethod
    def _check_offsets(offsets, npcm):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        nclips = len(offsets) - 1
        if nclips < 0 or (nclips > 0 and (offsets[0] < 0 or offsets[-1] > npcm or np.any(np.diff(offsets) < 0))):
            raise ValueError('offsets must be non-decreasing sample offsets inside pcm')
        return offsets, nclips

    def submit(self, pcm, offsets, want_hashes=True, want_peaks=False):
        """Queue a batch of host-resident PCM (float32, int16 or float64: ONE C-contiguous 1-D array, as extract(pcm=...) takes
        it) WITHOUT waiting for it: the copy to the device and the kernels run asynchronously when `pcm` is pinned host
        memory; call fetch(nclips, ...) for the result and keep `pcm` alive and unchanged until then.  With several
        Extractor contexts this overlaps the upload of one batch with the kernels of another."""
        pcm = np.asarray(pcm)
        if pcm.ndim != 1 or not pcm.flags.c_contiguous:
            # (no silent copy: the copy would be a temporary the asynchronous upload outlives)
            raise ValueError('submit: pcm must be one C-contiguous 1-D array (the upload is asynchronous: no copy is made)')
        offsets, nclips = self._check_offsets(offsets, pcm.size)
        flags = self._flags(want_hashes, want_peaks, False)
        self._submitted = (pcm, offsets)          # kept alive until the next submit / extract on this context
        self.last_nclips = nclips
        if pcm.dtype == np.int16:
            _lib.check(self.lib.afp_extract_host_s16(self.h, pcm.ctypes.data, offsets.ctypes.data, nclips, flags), 'afp_extract_host_s16')
        elif pcm.dtype == np.float32:
            _lib.check(self.lib.afp_extract_host(self.h, pcm.ctypes.data, offsets.ctypes.data, nclips, flags), 'afp_extract_host')
        elif pcm.dtype == np.float64:
            _lib.check(self.lib.afp_extract_host_f64(self.h, pcm.ctypes.data, offsets.ctypes.data, nclips, flags), 'afp_extract_host_f64')
        else:
            raise TypeError('submit: float32, int16 or float64 PCM')
        return nclips

    def extract_device(self, d_pcm_ptr, offsets, want_hashes=True, want_peaks=False, debug=False, s16=False):
        """Queue the hot path over PCM already resident in HBM (d_pcm_ptr = device address of the
        float32 -- or, with s16=True, int16 -- buffer `offsets` index into).  Results stay on the
        device; call fetch()."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self._last_offsets = offsets
        self.last_nclips = len(offsets) - 1
        flags = self._flags(want_hashes, want_peaks, debug)
        fn = self.lib.afp_extract_device_s16 if s16 else self.lib.afp_extract_device
        _lib.check(fn(self.h, C.c_void_p(int(d_pcm_ptr)), offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                      len(offsets) - 1, flags), 'afp_extract_device')

    def counts(self):
        th, tp, nu = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self.lib.afp_result_counts(self.h, C.byref(th), C.byref(tp), C.byref(nu)), 'afp_result_counts')
        return th.value, tp.value, nu.value

    def fetch(self, nclips, want_hashes=True, want_peaks=False):
        th, tp, nu = self.counts()
        r = BatchResult()
        r.nclips, r.shifts = nclips, self.shifts
        hp = op = pp = qp = None
        if want_hashes:
            r.hashes = np.empty((th, 2), dtype=np.int32)
            r.hash_offsets = np.zeros(nclips + 1, dtype=np.int64)
            hp, op = r.hashes.ctypes.data, r.hash_offsets.ctypes.data
        if want_peaks:
            r.peaks = np.empty((tp, 2), dtype=np.int32)
            r.peak_offsets = np.zeros(nu + 1, dtype=np.int64)
            pp, qp = r.peaks.ctypes.data, r.peak_offsets.ctypes.data
        r.unit_flags = np.zeros(nu, dtype=np.int32)
        _lib.check(self.lib.afp_fetch_all(self.h, hp, op, pp, qp, r.unit_flags.ctypes.data if nu else None), 'afp_fetch_all')
        return r

    def fetch_offsets(self, nclips):
        """(nclips + 1,) int64 row offsets per clip of the last extract, WITHOUT copying the rows (they stay in HBM for
        TableBuilder.store_batch): waits for the batch."""
        off = np.zeros(nclips + 1, dtype=np.int64)
        _lib.check(self.lib.afp_fetch_hashes(self.h, None, off.ctypes.data_as(C.POINTER(C.c_int64))), 'afp_fetch_hashes')
        return off

    # ---- pairing / hashing of given peak lists ------------------------------------------------
    def pairs_from_peaks(self, unit_peaks, want_hashes=True, want_landmarks=False):
        """unit_peaks: list (len = nclips*shifts, unit = clip*shifts + shift) of (P,2) arrays of
        (col, bin) rows as find_peaks / peaks_load produce them -- or in any list order Analyzer.peaks2landmarks accepts.  Returns (BatchResult with hashes
        per clip or None, list of (L,4) int32 landmark arrays per unit or None)."""
        nunits = len(unit_peaks)
        if nunits % self.shifts:
            raise ValueError('need one peak list per (clip, shift)')
        nclips = nunits // self.shifts
        arrs = [np.asarray(p, dtype=np.int32).reshape(-1, 2) for p in unit_peaks]
        for k, a in enumerate(arrs):
            # Analyzer.peaks2landmarks files the list into per-column lists in LIST order (audfprint_analyze.py:321-326) and
            # accepts any column order as long as the last row has the largest column.  Rows are stable-sorted by column
            # here (list order inside a column is kept).  Columns whose bins are ascending and unique -- what find_peaks
            # and peaks_load produce -- take the mask kernels; any other order (bins descending, a bin listed twice) is
            # paired from the rows in list order by k_pair_rows, exactly as the reference's nested loops do.
            if len(a) > 1 and np.any(np.diff(a[:, 0]) < 0):
                if int(a[-1, 0]) != int(a[:, 0].max()):
                    raise ValueError('peak list: the last row must hold the largest column (audfprint_analyze.py:321)')
                a = arrs[k] = a[np.argsort(a[:, 0], kind='stable')]
            if len(a) and (int(a[:, 0].max()) >= (1 << 24) or int(a[:, 0].min()) < 0):
                raise ValueError('peak list: column index outside 0 .. 2^24-1')
        upo = np.zeros(nunits + 1, dtype=np.int64)
        np.cumsum([len(a) for a in arrs], out=upo[1:])
        allp = np.ascontiguousarray(np.concatenate(arrs) if arrs else np.zeros((0, 2), np.int32), dtype=np.int32)
        flags = (_lib.WANT_HASHES if want_hashes else 0) | (_lib.WANT_LANDMARKS if want_landmarks else 0)
        I32, I64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        self.last_nclips = nclips
        _lib.check(self.lib.afp_pairs_from_peaks(self.h, allp.ctypes.data_as(I32), upo.ctypes.data_as(I64),
                                                 nclips, flags), 'afp_pairs_from_peaks')
        res, lms = None, None
        if want_hashes:
            th, _, _ = self.counts()
            res = BatchResult()
            res.nclips, res.shifts = nclips, self.shifts
            res.hashes = np.empty((th, 2), dtype=np.int32)
            res.hash_offsets = np.zeros(nclips + 1, dtype=np.int64)
            _lib.check(self.lib.afp_fetch_hashes(self.h, res.hashes.ctypes.data_as(I32),
                                                 res.hash_offsets.ctypes.data_as(I64)), 'afp_fetch_hashes')
        if want_landmarks:
            tot = C.c_int64()
            offs = np.zeros(nunits + 1, dtype=np.int64)
            _lib.check(self.lib.afp_fetch_landmarks(self.h, None, offs.ctypes.data_as(I64), C.byref(tot)), 'afp_fetch_landmarks')
            lm = np.empty((tot.value, 4), dtype=np.int32)
            _lib.check(self.lib.afp_fetch_landmarks(self.h, lm.ctypes.data_as(I32), offs.ctypes.data_as(I64), None), 'afp_fetch_landmarks')
            lms = [lm[offs[u]:offs[u + 1]] for u in range(nunits)]
        return res, lms

    def hashes_from_landmarks(self, landmarks):
        """(L,4) (time, bin1, bin2, dtime) -> (L,2) int32 (time, hash); audfprint_analyze.py:81-96."""
        lm = np.ascontiguousarray(np.asarray(landmarks, dtype=np.int32).reshape(-1, 4))
        out = np.zeros((lm.shape[0], 2), dtype=np.int32)
        if lm.shape[0]:
            I32 = C.POINTER(C.c_int32)
            _lib.check(self.lib.afp_hashes_from_landmarks(self.h, lm.ctypes.data_as(I32), lm.shape[0],
                                                          out.ctypes.data_as(I32)), 'afp_hashes_from_landmarks')
        return out

    def prune_spectrogram(self, sgram, a_dec, peaks=None, want_fwd=True, want_bwd=True):
        """The forward / backward threshold passes over a caller-supplied (256, T) spectrogram
        (audfprint_analyze.py:199-253).  Returns (fwd_mask, bwd_mask) as (256, T) uint8 arrays (None if not wanted)."""
        sg = np.asarray(sgram, dtype=np.float64)
        if sg.ndim != 2 or sg.shape[0] != 256:
            raise ValueError('audfprint_amd prunes 256-bin spectrograms only (n_fft = 512, Nyquist row dropped)')
        T = sg.shape[1]
        rows = np.ascontiguousarray(sg.T)
        U8 = C.POINTER(C.c_uint8)
        pk = None if peaks is None else np.ascontiguousarray((np.asarray(peaks).T != 0).astype(np.uint8))
        fwd = np.zeros((T, 256), np.uint8) if want_fwd else None
        bwd = np.zeros((T, 256), np.uint8) if want_bwd else None
        _lib.check(self.lib.afp_prune_spectrogram(self.h, rows.ctypes.data_as(C.POINTER(C.c_double)), T, float(a_dec),
                                                  None if pk is None else pk.ctypes.data_as(U8),
                                                  None if fwd is None else fwd.ctypes.data_as(U8),
                                                  None if bwd is None else bwd.ctypes.data_as(U8)), 'afp_prune_spectrogram')
        return (None if fwd is None else fwd.T, None if bwd is None else bwd.T)

    # ---- streams ------------------------------------------------------------------------------
    def set_pipeline(self, compact=None, compact_min_units=None, seg=None, seg_max_units=None, seg_len=None, seg_warm=None,
                     seg_force_fail=False, compact_force_timeout=False, hpf_force_fail=False):
        """Force / release the kernel path (afp_set_pipeline).  compact / seg: -1 the library's rule (by batch size), 0 off, 1 on;
        the other arguments positive values.  An argument left None takes the value the handle was CREATED with -- the
        library's defaults, or what AFP_COMPACT / AFP_SEG / AFP_COMPACT_MIN_UNITS / AFP_SEG_MAX_UNITS / AFP_SEG_LEN /
        AFP_SEG_WARM in the environment chose -- so set_pipeline() with no arguments undoes every earlier call and a handle
        configured through the environment stays configured that way.  Test hooks: seg_force_fail
        (afp_set_seg_force_fail: the segment-parallel scan's final check fails every unit; hpf_force_fail: the boundary check of
        the chunked onset filter fails instead), compact_force_timeout
        (afp_set_compact_force_timeout: one chunk of the compact stage withholds its state; the batch is re-run densely)."""
        def v(x, keep):
            return keep if x is None else int(x)
        _lib.check(self.lib.afp_set_pipeline(self.h, v(compact, -2), v(compact_min_units, 0), v(seg, -2), v(seg_max_units, 0),
                                             v(seg_len, 0), v(seg_warm, 0)), 'afp_set_pipeline')
        _lib.check(self.lib.afp_set_seg_force_fail(self.h, 2 if hpf_force_fail else 1 if seg_force_fail else 0), 'afp_set_seg_force_fail')
        _lib.check(self.lib.afp_set_compact_force_timeout(self.h, 1 if compact_force_timeout else 0), 'afp_set_compact_force_timeout')

    def path_stats(self):
        """Path of the batch last finalized (afp_get_path_stats): dict(compact, segments, redone_dense, redone_total, near_tie_units,
        near_tie_redone, near_tie_redone_total, hpf_chunked_total)."""
        out = (C.c_int32 * 8)()
        _lib.check(self.lib.afp_get_path_stats(self.h, out), 'afp_get_path_stats')
        return dict(compact=bool(out[0]), segments=bool(out[1]), redone_dense=bool(out[2]), redone_total=int(out[3]),
                    near_tie_units=int(out[4]), near_tie_redone=bool(out[5]), near_tie_redone_total=int(out[6]),
                    hpf_chunked_total=int(out[7]))

    def set_neartie_eps(self, eps=1e-11):
        """Near-tie guard of the threshold passes (afp_set_neartie_eps): units whose decisive comparisons came out closer than
        eps carry UNIT_NEARTIE; a compact-path batch that raised it is re-run on the dense path.  0 switches it off."""
        _lib.check(self.lib.afp_set_neartie_eps(self.h, float(eps)), 'afp_set_neartie_eps')

    def tie_frames(self):
        """(first, last) int32 arrays per unit: the frames whose non-zero samples all share one parity, above the floor (AFP_UNIT_TIE, include/afp.h)."""
        _, _, nu = self.counts()
        a, b = np.zeros(nu, np.int32), np.full(nu, -1, np.int32)
        if nu:
            I32 = C.POINTER(C.c_int32)
            _lib.check(self.lib.afp_fetch_unit_tie_frames(self.h, a.ctypes.data_as(I32), b.ctypes.data_as(I32)), 'afp_fetch_unit_tie_frames')
        return a, b

    def seg_stats(self):
        """Segment-parallel scan of the last batch (few long units): dict(used, segments, rerun_fwd, rerun_bwd, failed,
        failed_units, seg_len, seg_warm, short_cut_backoffs)."""
        out = (C.c_int32 * 8)()
        _lib.check(self.lib.afp_get_seg_stats(self.h, out), 'afp_get_seg_stats')
        return dict(used=bool(out[0]), segments=int(out[1]), rerun_fwd=int(out[2]), rerun_bwd=int(out[3]), failed=bool(out[4]),
                    failed_units=int(out[4]), seg_len=int(out[5]), seg_warm=int(out[6]), short_cut_backoffs=int(out[7]))

    def set_stream(self, hip_stream):
        """Run on an externally owned hipStream_t (int / None for the handle's own stream)."""
        _lib.check(self.lib.afp_set_stream(self.h, C.c_void_p(hip_stream or None)))

    def set_stage_streams(self, spectral, scan, pair=None):
        """Staged mode (afp_set_stage_streams): `spectral` / `scan` are raw hipStream_t values (ints, e.g.
        torch.cuda.Stream().cuda_stream) shared by all Extractors that should pipeline against each other;
        `pair` optionally gives the pairing kernels a stage of their own; (None, None) switches back."""
        _lib.check(self.lib.afp_set_stage_streams(self.h, C.c_void_p(spectral or None), C.c_void_p(scan or None),
                                                  C.c_void_p(pair or None)))

    def clock_probe_start(self, ms=5):
        """Start measuring the shader clock held while other work runs (afp_clock_probe_start)."""
        _lib.check(self.lib.afp_clock_probe_start(self.h, int(ms)), 'afp_clock_probe_start')

    def clock_probe_stop(self):
        mhz = C.c_double()
        _lib.check(self.lib.afp_clock_probe_stop(self.h, C.byref(mhz)), 'afp_clock_probe_stop')
        return round(mhz.value, 1)

    # ---- timing / debug ---------------------------------------------------------------------
    def set_timing(self, on):
        _lib.check(self.lib.afp_set_timing(self.h, 1 if on else 0))

    def reset_timings(self):
        _lib.check(self.lib.afp_reset_timings(self.h))

    def timings(self):
        ms = (C.c_double * _lib.AFP_NKERNELS)()
        n = (C.c_int64 * _lib.AFP_NKERNELS)()
        _lib.check(self.lib.afp_get_timings(self.h, ms, n))
        return {self.lib.afp_kernel_name(i).decode(): (ms[i], n[i]) for i in range(_lib.AFP_NKERNELS)}

    def debug(self, what, dtype, shape_tail=()):
        nbytes = self.lib.afp_debug_fetch(self.h, what, None, 0)
        _lib.check(nbytes, 'afp_debug_fetch')
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _lib.check(self.lib.afp_debug_fetch(self.h, what, out.ctypes.data_as(C.c_void_p), nbytes), 'afp_debug_fetch')
        return out.reshape((-1,) + tuple(shape_tail)) if shape_tail else out
