# coding=utf-8
"""Drop-in replacement for the reference module ``audfprint_analyze`` (dpwe/audfprint) whose hot
path -- STFT, log-spectrogram, decaying-threshold peak pick, peak pairing, hash packing, unique
sort -- runs as hand-written HIP kernels on an MI355X through libafp_hip.so (include/afp.h).

Use it by putting this package's directory ahead of the reference on ``sys.path`` under the name
``audfprint_analyze``, or ``import audfprint_amd.audfprint_analyze as
audfprint_analyze``; ``audfprint.py``, ``audfprint_match.py`` and the ``dpwe_*`` wrappers then run
unchanged.  The public surface mirrors the reference module (audfprint_analyze.py:31-78, 81-112,
115-457, 463-514): same names, arguments, return types, printed warnings and error behaviour.

There is NO CPU fallback: every method on the extraction path raises if the HIP library or a GPU
is missing.  The instance holds no device state (the per-process context lives in
``audfprint_amd.batch.Extractor``), so it pickles into joblib workers and survives fork into
``multiprocessing`` children exactly like the reference Analyzer (audfprint.py:218-224, 249-251).
"""
from __future__ import division, print_function

import os
import struct

import numpy as np

from .batch import Extractor
from . import _lib

# ############### Globals ############### #   (audfprint_analyze.py:29-33)
PRECOMPEXT = '.afpt'
PRECOMPPKEXT = '.afpk'

# Constants for Analyzer (audfprint_analyze.py:55-78)
DENSITY = 20.0
OVERSAMP = 1
N_FFT = 512
N_HOP = 256
HPF_POLE = 0.98
F1_BITS = 8
DF_BITS = 6
DT_BITS = 6
B1_MASK = (1 << F1_BITS) - 1
B1_SHIFT = DF_BITS + DT_BITS
DF_MASK = (1 << DF_BITS) - 1
DF_SHIFT = DT_BITS
DT_MASK = (1 << DT_BITS) - 1

# ---- which GPU this process uses (one process per GPU) ---------------------------------------------
# The reference's only parallel mechanism is `--ncores N`: N `multiprocessing.Process` children for new / add
# (audfprint.py:199-235) or N joblib workers for precompute / match (audfprint.py:243-267).  Every worker inherits
# the parent's environment, so an environment variable cannot tell them apart; what does is the ordinal their parent
# gave them when it started them (`multiprocessing.current_process()._identity`: (1,), (2,), ... for the children of
# multiproc_add and for loky's worker processes alike).  Worker k therefore opens GPU (k - 1) mod afp_device_count():
# `--ncores 8` on an 8-GPU node is one worker per GPU, `--ncores 16` two per GPU.  The process that started them --
# identity () -- takes GPU 0 and never has to initialise HIP to decide that (it forks right afterwards).
#   AFP_DEVICE=<n>      this GPU, whoever asks (explicit; inherited by every worker)
#   AFP_DEVICE=first    GPU 0 for every worker (the behaviour before round 6)
#   AFP_DEVICE=auto     (or unset) LOCAL_RANK when a launcher such as torchrun set it, else the worker ordinal rule above
#   AFP_DEVICE_COUNT=k  spread workers over the first k GPUs only (default: all that afp_device_count() reports)
_DEVICE_OF_PID = {}      # pid -> the choice this process made (a forked child inherits the dict, not the choice)


def _device_count():
    """GPUs the worker-ordinal rule spreads over: AFP_DEVICE_COUNT if set, else what the HIP runtime reports."""
    n = os.environ.get('AFP_DEVICE_COUNT', '').strip()
    if n:
        return max(1, int(n))
    return max(1, int(_lib.load().afp_device_count()))


def _worker_ordinal():
    """0 for a process nobody started as a worker; k (1-based) for the k-th process its parent started -- the
    children of audfprint.multiproc_add (audfprint.py:217-224) and joblib's loky workers (audfprint.py:249, 259) --
    and the pid for a multiprocessing child that carries no identity."""
    import multiprocessing
    ident = getattr(multiprocessing.current_process(), '_identity', None)
    if ident:
        return int(ident[-1])
    parent = getattr(multiprocessing, 'parent_process', lambda: None)()
    return os.getpid() if parent is not None else 0


def _device():
    pid = os.getpid()
    dev = _DEVICE_OF_PID.get(pid)
    if dev is None:
        spec = os.environ.get('AFP_DEVICE', 'auto').strip().lower() or 'auto'
        if spec == 'first':
            dev = 0
        elif spec != 'auto':
            dev = int(spec)
        elif os.environ.get('LOCAL_RANK', '').strip():
            dev = int(os.environ['LOCAL_RANK'])
        else:
            k = _worker_ordinal()
            dev = 0 if k == 0 else (k - 1) % _device_count()
        _DEVICE_OF_PID.clear()                   # (entries of other pids belong to ancestors of a fork)
        _DEVICE_OF_PID[pid] = dev
    return dev


def locmax(vec, indices=False):
    """Boolean vector of local maxima (>= on the left, strict on the right, end points allowed).
    Host-side helper kept for API parity (audfprint_analyze.py:36-52); the extraction path does
    this inside k_scan."""
    vec = np.asarray(vec)
    nbr = np.zeros(len(vec) + 1, dtype=bool)
    nbr[0] = True
    nbr[1:-1] = np.greater_equal(vec[1:], vec[:-1])
    maxmask = (nbr[:-1] & ~nbr[1:])
    if indices:
        return np.nonzero(maxmask)[0]
    return maxmask


def landmarks2hashes(landmarks):
    """(time, bin1, bin2, dtime) landmarks -> (N,2) int32 (time, hash); audfprint_analyze.py:81-96.
    Packed on the GPU (k_lm2hash)."""
    landmarks = np.array(landmarks)
    if landmarks.shape[0] == 0:
        return np.zeros((0, 2), dtype=np.int32)
    return Extractor.get(_device()).hashes_from_landmarks(landmarks)


def hashes2landmarks(hashes):
    """Inverse of landmarks2hashes; audfprint_analyze.py:99-112 (query-side display helper)."""
    landmarks = []
    for time_, hash_ in hashes:
        dtime = hash_ & DT_MASK
        bin1 = (hash_ >> B1_SHIFT) & B1_MASK
        dbin = (hash_ >> DF_SHIFT) & DF_MASK
        if dbin >= (1 << (DF_BITS - 1)):
            dbin -= (1 << DF_BITS)
        landmarks.append((time_, bin1, bin1 + dbin, dtime))
    return landmarks


class Analyzer(object):
    """Parameter bag + per-file methods of the reference Analyzer (audfprint_analyze.py:115-457),
    re-hosted on the HIP library.  Attributes are read at call time: audfprint.py mutates them
    after construction (audfprint.py:285-298)."""

    def __init__(self, density=DENSITY):
        self.density = density
        self.target_sr = 11025
        self.n_fft = N_FFT
        self.n_hop = N_HOP
        self.shifts = 1
        self.f_sd = 30.0
        self.maxpksperframe = 5
        self.maxpairsperpeak = 3
        self.targetdf = 31
        self.mindt = 2
        self.targetdt = 63
        self.soundfiledur = 0.0
        self.soundfiletotaldur = 0.0
        self.soundfilecount = 0
        self.fail_on_error = True

    # ---- device context -----------------------------------------------------------------------
    def _extractor(self, shifts):
        ex = Extractor.get(_device())
        ex.set_params(density=self.density, maxpksperframe=self.maxpksperframe,
                      maxpairsperpeak=self.maxpairsperpeak, f_sd=self.f_sd, shifts=shifts,
                      targetdf=self.targetdf, mindt=self.mindt, targetdt=self.targetdt,
                      n_fft=self.n_fft, n_hop=self.n_hop)
        return ex

    @staticmethod
    def _as_pcm(d):
        """float32 (what audio_read produces, audio_read.py:76) goes to the GPU as it is; anything else is carried
        as float64 with the same values -- the reference multiplies whatever dtype it is given by the float64
        window (stft.py:93), so a float64 waveform must not be rounded to float32 on the way in (and an integer
        array is NOT rescaled by 1/32768: that is audio_read's job, and the batch API's s16 entry)."""
        d = np.asarray(d)
        if d.ndim != 1:
            d = d.reshape(-1)
        if d.dtype == np.float32:
            return np.ascontiguousarray(d)
        d = np.ascontiguousarray(d, dtype=np.float64)
        # The kernels form |S|^2 before the log (the reference takes np.abs of the complex bins, a hypot): a float64
        # waveform below ~1e-150 or above ~1e+150 would under/overflow there.  The path is invariant to a
        # power-of-two gain (exact in the FFT; log|S| shifts by a constant that the mean subtraction removes,
        # audfprint_analyze.py:285-286), so such a waveform is brought to unit scale first.
        if d.size:
            amax = float(np.max(np.abs(d)))
            if np.isfinite(amax) and amax != 0.0 and not (2.0 ** -300 < amax < 2.0 ** 300):
                d = np.ldexp(d, -int(np.frexp(amax)[1]))
        return d

    @staticmethod
    def _warn_zero(flags):
        for f in flags:
            if f & _lib.UNIT_ZERO:
                # audfprint_analyze.py:290, once per find_peaks call
                print("find_peaks: Warning: input signal is identically zero.")
            if f & _lib.UNIT_TIE:
                # no counterpart in the reference: a frame whose non-zero samples all sit at even (or all at odd) offsets
                # has bins that are equal in exact arithmetic, and which of them the reference picks is decided by the
                # rounding noise of numpy's FFT (include/afp.h, AFP_UNIT_TIE)
                import warnings
                warnings.warn("audfprint_amd: a sparse frame (all non-zero samples at offsets of one parity, e.g. a lone "
                              "click in digital silence); the peaks picked in it are decided by FFT rounding noise and "
                              "may differ from numpy's", RuntimeWarning, stacklevel=3)

    # ---- host-side helpers with the reference's names (the extraction path does this inside k_scan) --------
    def spreadpeaks(self, peaks, npoints=None, width=4.0, base=None):
        """Element-wise max of Gaussian bumps val * exp(-0.5 ((k - pos) / width)^2) over ``base`` (or zeros of
        length npoints); audfprint_analyze.py:162-197."""
        vec = np.zeros(npoints) if base is None else np.copy(base)
        k = np.arange(len(vec))
        table = np.exp(-0.5 * ((np.arange(-len(vec), len(vec) + 1) / width) ** 2))     # same expression as :191-192
        for pos, val in peaks:
            vec = np.maximum(vec, val * table[k + len(vec) - pos])
        return vec

    def spreadpeaksinvector(self, vector, width=4.0):
        """spreadpeaks over every local maximum of ``vector``; audfprint_analyze.py:153-160."""
        vector = np.asarray(vector)
        idx = locmax(vector, indices=True)
        return self.spreadpeaks(zip(idx, vector[idx]), npoints=len(vector), width=width)

    # ---- the two passes of the peak picker over a caller-supplied spectrogram (k_scan, raw-row mode) ----
    def _decaying_threshold_fwd_prune(self, sgram, a_dec):
        """Forward pass of find_peaks over a (256, T) onset-filtered spectrogram: float 0/1 array of the same shape
        with the (at most maxpksperframe) local maxima per column that exceed the decaying threshold;
        audfprint_analyze.py:199-231."""
        ex = self._extractor(1)
        fwd, _ = ex.prune_spectrogram(sgram, a_dec, want_fwd=True, want_bwd=False)
        return fwd.astype(np.float64)

    def _decaying_threshold_bwd_prune_peaks(self, sgram, peaks, a_dec):
        """Backward pass: prunes `peaks` (modified in place and returned, like the reference does);
        audfprint_analyze.py:233-253."""
        ex = self._extractor(1)
        _, bwd = ex.prune_spectrogram(sgram, a_dec, peaks=peaks, want_fwd=False, want_bwd=True)
        if isinstance(peaks, np.ndarray):
            peaks[...] = bwd
            return peaks
        return bwd.astype(np.float64)

    # ---- the hot path ---------------------------------------------------------------------------
    def find_peaks(self, d, sr):
        """Local peaks of the spectrogram: list of (time_frame, freq_bin); audfprint_analyze.py:255-308."""
        if len(d) == 0:
            return []
        ex = self._extractor(1)
        r = ex.extract(clips=[self._as_pcm(d)], want_hashes=False, want_peaks=True)
        self._warn_zero(r.unit_flags)
        return list(map(tuple, r.unit_peaks(0, 0).tolist()))          # (col, bin) tuples of Python ints

    def peaks2landmarks(self, pklist):
        """(col, bin) peaks -> list of (col, peak, peak2, col2-col); audfprint_analyze.py:310-343."""
        if len(pklist) == 0:
            return []
        ex = self._extractor(1)
        _, lms = ex.pairs_from_peaks([np.asarray(pklist, dtype=np.int32).reshape(-1, 2)],
                                     want_hashes=False, want_landmarks=True)
        return list(map(tuple, lms[0].tolist()))                      # (col, f1, f2, dt) tuples of Python ints

    def _read_audio(self, filename):
        """The audio_read call and error convention of wavfile2peaks (audfprint_analyze.py:356-368)."""
        import audio_read  # the reference's decoder module (ffmpeg pipe / wav), reused unchanged
        try:
            d, sr = audio_read.audio_read(filename, sr=self.target_sr, channels=1)
        except Exception as e:  # audioread.NoBackendError:
            message = "wavfile2peaks: Error reading " + filename
            if self.fail_on_error:
                print(e)
                raise IOError(message)
            print(message, "skipping")
            d = []
            sr = self.target_sr
        return d, sr

    def _account(self, dur):
        self.soundfiledur = dur
        self.soundfiletotaldur += dur
        self.soundfilecount += 1

    def wavfile2peaks(self, filename, shifts=None):
        """Landmark peaks of a soundfile: list of (time, bin), or a list of such lists when
        shifts >= 2; audfprint_analyze.py:345-383."""
        ext = os.path.splitext(filename)[1]
        if ext == PRECOMPPKEXT:
            peaks = peaks_load(filename)
            dur = np.max(peaks, axis=0)[0] * self.n_hop / self.target_sr
        else:
            d, sr = self._read_audio(filename)
            dur = len(d) / sr
            if shifts is None or shifts < 2:
                peaks = self.find_peaks(d, sr)
            else:
                # all part-frame shifts in one batch; the sample offsets follow self.shifts exactly
                # like the reference loop (:374-376: int(shift / self.shifts * self.n_hop))
                pcm = self._as_pcm(d)
                offs = [int(shift / self.shifts * self.n_hop) for shift in range(shifts)]
                clips = [pcm[o:] for o in offs]
                ex = self._extractor(1)
                r = ex.extract(clips=clips, want_hashes=False, want_peaks=True)
                self._warn_zero(r.unit_flags)
                peaks = [list(map(tuple, r.unit_peaks(i, 0).tolist())) for i in range(shifts)]
        self._account(dur)
        return peaks

    def wavfile2hashes(self, filename):
        """Fingerprint hashes of a soundfile as an (N,2) int32 array of sorted unique
        (time, hash) rows; audfprint_analyze.py:385-426.  Audio input goes through ONE fused GPU
        pipeline (peaks never leave the device); '.afpk' peak files are paired/hashed on the GPU;
        '.afpt' hash files are simply loaded."""
        ext = os.path.splitext(filename)[1]
        if ext == PRECOMPEXT:
            hashes = hashes_load(filename)
            dur = np.max(hashes, axis=0)[0] * self.n_hop / self.target_sr
            self._account(dur)
            return hashes
        if ext == PRECOMPPKEXT:
            peaks = self.wavfile2peaks(filename, self.shifts)
            if len(peaks) == 0:
                return []
            ex = self._extractor(1)
            res, _ = ex.pairs_from_peaks([np.asarray(peaks, dtype=np.int32).reshape(-1, 2)], want_hashes=True)
            return res.clip_hashes(0)
        d, sr = self._read_audio(filename)
        dur = len(d) / sr
        self._account(dur)
        multi = not (self.shifts is None or self.shifts < 2)
        if len(d) == 0:
            # find_peaks returns [] per shift (:273-274): [] for one shift (:401-402), an empty
            # (0,2) array through the concatenate/unique path for several (:404-422)
            return np.zeros((0, 2), dtype=np.int32) if multi else []
        ex = self._extractor(self.shifts)
        pcm = self._as_pcm(d)
        r = ex.extract(clips=[pcm], want_hashes=True, want_peaks=False)
        self._warn_zero(r.unit_flags)
        hashes = r.clip_hashes(0)
        if not multi and len(hashes) == 0:
            # No hashes: the reference returns [] when there was no PEAK at all (:401-402) and an empty (0,2) array when
            # peaks exist but pair into nothing.  Only this rare case asks the device for the peak list (the common one
            # saves its two launches and the copy: 0.04 ms of a 10 s file's 0.44 ms).
            rp = ex.extract(clips=[pcm], want_hashes=False, want_peaks=True)
            if len(rp.unit_peaks(0, 0)) == 0:
                return []                                               # :401-402
        return hashes

    # ---- bulk forms (no counterpart in the reference, whose CLI feeds one file at a time: audfprint.py:164-165, 177-182) ----
    def _many(self, filenames):
        """(results, durations) of the files in list order; see wavfiles2hashes."""
        filenames = list(filenames)
        out = [None] * len(filenames)
        clips, where, durs = [], [], [None] * len(filenames)
        multi = not (self.shifts is None or self.shifts < 2)
        for i, fn in enumerate(filenames):
            if os.path.splitext(fn)[1] in (PRECOMPEXT, PRECOMPPKEXT):
                continue                                               # precomputed files: per file below, in list order
            d, sr = self._read_audio(fn)
            durs[i] = len(d) / sr
            if len(d) == 0:
                out[i] = np.zeros((0, 2), dtype=np.int32) if multi else []          # :273-274, :401-402, :404-422
            else:
                where.append(i)
                clips.append(self._as_pcm(d))
        if clips:
            ex = self._extractor(self.shifts)
            r = ex.extract(clips=clips, want_hashes=True, want_peaks=False)
            self._warn_zero(r.unit_flags)
            empty = [k for k in range(len(clips)) if not multi and r.hash_offsets[k + 1] == r.hash_offsets[k]]
            nopeak = set()
            if empty:
                # no rows: [] when there was no PEAK at all (:401-402), an empty array when peaks pair into nothing -- only
                # these clips ask the device for their peak lists
                rp = ex.extract(clips=[clips[k] for k in empty], want_hashes=False, want_peaks=True)
                nopeak = set(k for j, k in enumerate(empty) if len(rp.unit_peaks(j, 0)) == 0)
            for k, i in enumerate(where):
                out[i] = [] if k in nopeak else r.clip_hashes(k).copy()
        # bookkeeping, and the precomputed files, in list order: what a loop over wavfile2hashes leaves behind
        for i, fn in enumerate(filenames):
            if durs[i] is None:
                out[i] = self.wavfile2hashes(fn)
                durs[i] = self.soundfiledur
            else:
                self._account(durs[i])
        return out, durs

    def wavfiles2hashes(self, filenames):
        """``[self.wavfile2hashes(f) for f in filenames]`` -- the same list, element for element (an (N,2) int32 array per
        audio file, ``[]`` where a single-shift analysis finds no peak, a python list for an '.afpt' file), the same
        ``soundfiledur / soundfiletotaldur / soundfilecount`` bookkeeping, the same warnings and error convention -- with ONE
        launch of the hot path for all the audio files instead of one per file (the batch API behind the Analyzer's own
        parameters).  Files are decoded by the caller's ``audio_read`` module one after another, as ``wavfile2hashes`` does."""
        return self._many(filenames)[0]

    def ingest_many(self, hashtable, filenames):
        """``[self.ingest(hashtable, f) for f in filenames]`` with one launch for the extraction and the reference's own
        ``hashtable.store`` per file, in list order (audfprint_analyze.py:452-453); returns the list of (dur, nhashes).  For a
        whole job on a device-resident table use audfprint_amd.table.TableBuilder."""
        filenames = list(filenames)
        hashes, durs = self._many(filenames)
        res = []
        for fn, h, dur in zip(filenames, hashes, durs):
            hashtable.store(fn, h)
            res.append((dur, len(h)))
        return res

    # ########## functions to link to actual hash table index database ###### #
    def ingest(self, hashtable, filename):
        """Read an audio file and add it to the database; returns (dur, nhashes);
        audfprint_analyze.py:430-457."""
        hashes = self.wavfile2hashes(filename)
        hashtable.store(filename, hashes)
        return self.soundfiledur, len(hashes)


# ########## functions to read/write hashes to file for a single track #### #
# (audfprint_analyze.py:463-514; same bytes: 16-byte magic + little-endian <2i records)
HASH_FMT = '<2i'
HASH_MAGIC = b'audfprinthashV00'
PEAK_FMT = '<2i'
PEAK_MAGIC = b'audfprintpeakV00'


def _save_pairs(filename, magic, pairs):
    arr = np.asarray(pairs)
    with open(filename, 'wb') as f:
        f.write(magic)
        if arr.size:
            f.write(np.ascontiguousarray(arr.reshape(-1, 2)).astype('<i4').tobytes())


def _load_pairs(filename, magic, what):
    fmtsize = struct.calcsize(HASH_FMT)
    with open(filename, 'rb') as f:
        got = f.read(len(magic))
        if got != magic:
            raise IOError('%s is not a %s file (magic %s)' % (filename, what, got))
        data = f.read()
    n = len(data) // fmtsize
    arr = np.frombuffer(data[:n * fmtsize], dtype='<i4').reshape(-1, 2)
    return [(int(a), int(b)) for a, b in arr]


def hashes_save(hashfilename, hashes):
    """Write (time, hash) pairs as 32 bit ints; audfprint_analyze.py:469-474."""
    _save_pairs(hashfilename, HASH_MAGIC, hashes)


def hashes_load(hashfilename):
    """Read back hashes written by hashes_save (list of tuples); audfprint_analyze.py:477-490."""
    return _load_pairs(hashfilename, HASH_MAGIC, 'hash')


def peaks_save(peakfilename, peaks):
    """Write (time, bin) pairs as 32 bit ints; audfprint_analyze.py:493-498."""
    _save_pairs(peakfilename, PEAK_MAGIC, peaks)


def peaks_load(peakfilename):
    """Read back (time, bin) pairs written by peaks_save; audfprint_analyze.py:501-514."""
    return _load_pairs(peakfilename, PEAK_MAGIC, 'peak')


# ####### legacy entry points other modules import (audfprint_match.py:477-479 uses glob2hashtable /
# g2h_analyzer; extract_features is the Gordon hook, audfprint_analyze.py:520-553).  Thin delegations to the
# Analyzer above; the reference's ad-hoc `local_tester` is not part of the interface and is not provided.
extract_features_analyzer = None
g2h_analyzer = None


def extract_features(track_obj, *args, **kwargs):
    """Hashes of ``track_obj.fn_audio`` with density / n_fft / n_hop / sr taken from the keyword arguments
    (module defaults where one is absent)."""
    global extract_features_analyzer
    an = extract_features_analyzer = extract_features_analyzer or Analyzer()
    for attr, key, default in (('density', 'density', DENSITY), ('n_fft', 'n_fft', N_FFT), ('n_hop', 'n_hop', N_HOP),
                               ('target_sr', 'sr', 11025)):
        val = kwargs.get(key)
        setattr(an, attr, default if val is None else val)
    return an.wavfile2hashes(track_obj.fn_audio)


def glob2hashtable(pattern, density=20.0):
    """A new ``hash_table.HashTable`` (the caller's own module) holding every file that matches ``pattern``."""
    import glob
    import hash_table
    global g2h_analyzer
    g2h_analyzer = g2h_analyzer or Analyzer(density=density)
    table = hash_table.HashTable()
    for name in glob.glob(pattern):
        g2h_analyzer.ingest(table, name)
    return table
