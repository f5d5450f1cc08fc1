"""CPU ORACLE for the audfprint landmark-extraction hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm (dpwe/audfprint) for the
path  PCM -> STFT -> log-spectrogram -> HPF -> decaying-threshold peak pick ->
peak pairs -> 20-bit hashes -> sorted unique (time, hash).  It exists so the HIP path
can be checked; it is NOT part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``audfprint_amd``) never imports it and fails loudly if its HIP
extension is missing.

Parity pinning: the reference has no golden vectors of its own, so this
restatement is pinned against (a) the live reference imported from /root/reference in
the build container (tests/test_oracle_vs_reference.py, skipped where the tree is
absent) and (b) frozen fixtures generated from the live reference by
tests/golden/make_golden.py (tests/golden/*.npz).  Integer outputs (peaks, landmarks,
hashes) equal the reference bit-for-bit on every fixture; float intermediates equal
it to the last bit as well because the same numpy/scipy primitives are used in the
same order (np.fft.rfft, np.log, np.mean) except the HPF, which restates
scipy.signal.lfilter's direct-form-II-transposed recurrence.

Every function cites the reference file:line it follows (paths relative to the
reference repository root).
"""
from __future__ import annotations

import numpy as np

# ---- constants: audfprint_analyze.py:55-78 -------------------------------------------------
N_FFT = 512
N_HOP = 256
HPF_POLE = 0.98
OVERSAMP = 1
F1_BITS, DF_BITS, DT_BITS = 8, 6, 6
B1_MASK = (1 << F1_BITS) - 1
B1_SHIFT = DF_BITS + DT_BITS
DF_MASK = (1 << DF_BITS) - 1
DF_SHIFT = DT_BITS
DT_MASK = (1 << DT_BITS) - 1


class Params:
    """Parameter bag = the Analyzer attributes that steer the path
    (audfprint_analyze.py:125-151)."""

    def __init__(self, density=20.0, maxpksperframe=5, maxpairsperpeak=3, f_sd=30.0,
                 shifts=1, targetdf=31, mindt=2, targetdt=63, n_fft=N_FFT, n_hop=N_HOP):
        self.density = density
        self.maxpksperframe = maxpksperframe
        self.maxpairsperpeak = maxpairsperpeak
        self.f_sd = f_sd
        self.shifts = shifts
        self.targetdf = targetdf
        self.mindt = mindt
        self.targetdt = targetdt
        self.n_fft = n_fft
        self.n_hop = n_hop


def a_dec_of(density, n_hop=N_HOP):
    """Masking-envelope decay constant, audfprint_analyze.py:277."""
    return (1 - 0.01 * (density * np.sqrt(n_hop / 352.8) / 35)) ** (1 / OVERSAMP)


def hann_window(n_fft=N_FFT):
    """np.hanning(n_fft + 2)[1:-1] -- the symmetric Hann stripped of its zero end
    points, audfprint_analyze.py:279."""
    return np.hanning(n_fft + 2)[1:-1]


def gauss_table(npoints=256, width=30.0):
    """The cached Gaussian profile __sp_vals, audfprint_analyze.py:191-192.
    Index k+npoints holds exp(-0.5*(k/width)^2) for k in [-npoints, npoints]."""
    return np.exp(-0.5 * ((np.arange(-npoints, npoints + 1) / width) ** 2))


# ---- row 1: stft.py:62-94 -------------------------------------------------------------------
def reflect_index(i, n):
    """Source index in d[0:n] of padded index i-256... generalised numpy 'reflect'
    (np.pad(signal, n_fft//2, mode='reflect'), stft.py:87-88): a mirrored periodic
    extension with period 2(n-1); for n == 1 every index maps to 0."""
    i = np.asarray(i, dtype=np.int64)
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    m = np.mod(i, period)
    return np.where(m >= n, period - m, m)


def stft_complex(d, n_fft=N_FFT, n_hop=N_HOP, window=None):
    """stft.stft (stft.py:62-94) with the window array of audfprint_analyze.py:279.
    Returns complex128 (n_fft/2+1, T), T = 1 + len(d)//n_hop."""
    d = np.asarray(d)
    if window is None:
        window = hann_window(n_fft)
    x = np.pad(d, n_fft // 2, mode='reflect')                      # stft.py:88
    nfr = 1 + (x.shape[0] - n_fft) // n_hop                         # stft.py:33
    idx = (np.arange(nfr)[:, None] * n_hop) + np.arange(n_fft)[None, :]
    frames = x[idx]                                                 # stft.py:32-36 (copy of the strided view)
    windowed = frames * window                                      # stft.py:93 (f32*f64 -> f64)
    return np.fft.rfft(windowed, n_fft).transpose()                 # stft.py:94


# ---- row 2: audfprint_analyze.py:280-295 ------------------------------------------------------
def log_sgram(S):
    """abs / max / log(max(S, max/1e6)) / subtract mean; audfprint_analyze.py:280-290.
    Returns (sgram (257,T) float64, is_zero flag)."""
    sgram = np.abs(S)
    sgrammax = np.max(sgram)
    if sgrammax > 0.0:
        sgram = np.log(np.maximum(sgram, np.max(sgram) / 1e6))
        sgram = sgram - np.mean(sgram)
        return sgram, False
    return sgram, True


def hpf_rows(sgram):
    """scipy.signal.lfilter([1,-1],[1,-0.98]) along time for every row, then drop the
    Nyquist row (audfprint_analyze.py:293-295).  Restated as lfilter's direct-form-II
    transposed recurrence:  y[n] = x[n] + z ;  z = -x[n] + 0.98*y[n]  (z starts at 0)."""
    pole = HPF_POLE ** (1 / OVERSAMP)
    x = np.ascontiguousarray(sgram[:-1, :])
    y = np.empty_like(x)
    z = np.zeros(x.shape[0])
    for n in range(x.shape[1]):
        xn = x[:, n]
        yn = xn + z
        z = xn * (-1.0) - yn * (-pole)
        y[:, n] = yn
    return y


# ---- row 3: audfprint_analyze.py:36-52 --------------------------------------------------------
def locmax(vec):
    """Boolean local-max mask: >= on the left, strict on the right, end points allowed."""
    nbr = np.zeros(len(vec) + 1, dtype=bool)
    nbr[0] = True
    nbr[1:-1] = np.greater_equal(vec[1:], vec[:-1])
    return nbr[:-1] & ~nbr[1:]


# ---- row 4: audfprint_analyze.py:153-197 ------------------------------------------------------
def spreadpeaks(peaks, G, base=None, npoints=256):
    """vec = max(vec, val*G[i + n - pos]) for each (pos, val); audfprint_analyze.py:162-197."""
    vec = np.zeros(npoints) if base is None else np.copy(base)
    n = len(vec)
    for pos, val in peaks:
        vec = np.maximum(vec, val * G[n - pos: 2 * n - pos])
    return vec


def spreadpeaksinvector(vector, G):
    """Spread every local max of `vector`; audfprint_analyze.py:153-160."""
    pk = np.nonzero(locmax(vector))[0]
    return spreadpeaks(zip(pk, vector[pk]), G, npoints=len(vector))


# ---- row 5: audfprint_analyze.py:199-231 ------------------------------------------------------
def fwd_prune(sgram, a_dec, G, maxpksperframe):
    """Forward decaying-threshold pass.  Returns (mask (256,T) float, list per column of
    the accepted (val, bin) in the reference's descending order)."""
    srows, scols = sgram.shape
    sthresh = spreadpeaksinvector(np.max(sgram[:, :min(10, scols)], axis=1), G)
    peaks = np.zeros((srows, scols))
    for col in range(scols):
        s_col = sgram[:, col]
        pos = np.nonzero(locmax(s_col) & (s_col > sthresh))[0]
        valspeaks = sorted(zip(s_col[pos], pos), reverse=True)
        for val, p in valspeaks[:maxpksperframe]:
            sthresh = np.maximum(sthresh, val * G[srows - p: 2 * srows - p])
            peaks[p, col] = 1
        sthresh = sthresh * a_dec
    return peaks


# ---- row 6: audfprint_analyze.py:233-253 ------------------------------------------------------
def bwd_prune(sgram, peaks, a_dec, G):
    """Backward pass; mutates and returns the mask."""
    srows, scols = sgram.shape
    sthresh = spreadpeaksinvector(sgram[:, -1], G)
    for col in range(scols, 0, -1):
        pkposs = np.nonzero(peaks[:, col - 1])[0]
        peakvals = sgram[pkposs, col - 1]
        for val, p in sorted(zip(peakvals, pkposs), reverse=True):
            if val >= sthresh[p]:
                sthresh = np.maximum(sthresh, val * G[srows - p: 2 * srows - p])
                if col < scols:
                    peaks[p, col] = 0
            else:
                peaks[p, col - 1] = 0
        sthresh = a_dec * sthresh
    return peaks


# ---- rows 1-7 together: audfprint_analyze.py:255-308 -------------------------------------------
def find_peaks_stages(d, prm=None):
    """All intermediate stages of Analyzer.find_peaks for one clip, as a dict:
    mag (257,T), logs (257,T, mean-subtracted), sgram (256,T, HPF'd), fwd (256,T mask),
    peaks (P,2) int32 [(col, bin)], zero (bool)."""
    prm = prm or Params()
    d = np.asarray(d)
    out = {}
    if len(d) == 0:                                                 # :273-274
        out['peaks'] = np.zeros((0, 2), np.int32)
        return out
    a_dec = a_dec_of(prm.density, prm.n_hop)
    S = stft_complex(d, prm.n_fft, prm.n_hop)
    out['mag'] = np.abs(S)
    logs, zero = log_sgram(S)
    out['zero'] = zero
    out['logs'] = logs
    sgram = hpf_rows(logs)
    out['sgram'] = sgram
    G = gauss_table(sgram.shape[0], prm.f_sd)
    fwd = fwd_prune(sgram, a_dec, G, prm.maxpksperframe)
    out['fwd'] = fwd.copy()
    msk = bwd_prune(sgram, fwd, a_dec, G)
    cols, bins = np.nonzero(msk.T)                                  # col-major, bin ascending (:303-308)
    out['peaks'] = np.stack([cols, bins], axis=1).astype(np.int32)
    return out


def find_peaks(d, prm=None):
    """Analyzer.find_peaks: (P,2) int32 array of (col, bin)."""
    return find_peaks_stages(d, prm)['peaks']


# ---- row 8: audfprint_analyze.py:310-343 ------------------------------------------------------
def peaks2landmarks(peaks, prm=None):
    """(P,2) (col,bin) -> (L,4) int64 (col, f1, f2, dt) in the reference's nested order."""
    prm = prm or Params()
    peaks = np.asarray(peaks).reshape(-1, 2)
    lm = []
    if len(peaks) > 0:
        scols = int(peaks[-1][0]) + 1
        peaks_at = [[] for _ in range(scols)]
        for col, b in peaks:
            peaks_at[int(col)].append(int(b))
        for col in range(scols):
            for peak in peaks_at[col]:
                n = 0
                for col2 in range(col + prm.mindt, min(scols, col + prm.targetdt)):
                    if n >= prm.maxpairsperpeak:
                        break
                    for peak2 in peaks_at[col2]:
                        if abs(peak2 - peak) < prm.targetdf and n < prm.maxpairsperpeak:
                            lm.append((col, peak, peak2, col2 - col))
                            n += 1
    return np.array(lm, dtype=np.int64).reshape(-1, 4)


# ---- row 9: audfprint_analyze.py:81-96 --------------------------------------------------------
def landmarks2hashes(lm):
    lm = np.asarray(lm, dtype=np.int64).reshape(-1, 4)
    h = np.zeros((lm.shape[0], 2), dtype=np.int32)
    if lm.shape[0] == 0:
        return h
    h[:, 0] = lm[:, 0]
    h[:, 1] = (((lm[:, 1] & B1_MASK) << B1_SHIFT)
               | (((lm[:, 2] - lm[:, 1]) & DF_MASK) << DF_SHIFT)
               | (lm[:, 3] & DT_MASK))
    return h


# ---- row 10: audfprint_analyze.py:414-422 -----------------------------------------------------
def unique_sort_hashes(h):
    key = (h[:, 0].astype(np.uint64) << np.uint64(32)) + h[:, 1].astype(np.uint64)
    u = np.sort(np.unique(key))
    return np.hstack([(u >> np.uint64(32))[:, None],
                      (u & np.uint64((1 << 32) - 1))[:, None]]).astype(np.int32)


# ---- row 11 + wavfile2hashes glue: audfprint_analyze.py:369-377, 400-422 -----------------------
def shift_offsets(shifts, n_hop=N_HOP):
    """Sample offsets of the part-frame shifts, audfprint_analyze.py:374-376."""
    if shifts is None or shifts < 2:
        return [0]
    return [int(s / shifts * n_hop) for s in range(shifts)]


def extract(d, prm=None):
    """Whole path for one clip.  Returns (peaklists, hashes): one (P,2) int32 peak array
    per shift, and the (N,2) int32 sorted-unique (time, hash) array (empty (0,2) when
    there are no peaks -- the reference returns [] there, audfprint_analyze.py:401-402)."""
    prm = prm or Params()
    d = np.asarray(d)
    peaklists = [find_peaks(d[off:], prm) for off in shift_offsets(prm.shifts, prm.n_hop)]
    hs = [landmarks2hashes(peaks2landmarks(p, prm)) for p in peaklists]
    allh = np.concatenate(hs) if hs else np.zeros((0, 2), np.int32)
    if allh.shape[0] == 0:
        return peaklists, np.zeros((0, 2), np.int32)
    return peaklists, unique_sort_hashes(allh)


# ---- synthetic input recipe (identical to audio_read.buf_to_float :121-145) ------
def synth_noise(seed, secs, sr=11025, nsamp=None):
    rng = np.random.RandomState(seed)
    n = int(round(sr * secs)) if nsamp is None else nsamp
    x = rng.randn(n) * 0.1
    pcm = np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    return pcm.astype(np.float32) / np.float32(32768)


def synth_tonal(seed, secs, sr=11025):
    """12 FM sinusoids x 2 Hz square gate + 0.001 noise, int16-quantised: produces
    digital-silence plateaus (the class where float32 flips peaks)."""
    rng = np.random.RandomState(seed)
    n = int(round(sr * secs))
    t = np.arange(n) / sr
    x = np.zeros(n)
    for _ in range(12):
        f0 = rng.uniform(200, 4500)
        fm = rng.uniform(0.5, 6.0)
        dev = rng.uniform(0, 40)
        ph = rng.uniform(0, 2 * np.pi)
        x += rng.uniform(0.02, 0.08) * np.sin(2 * np.pi * f0 * t + dev / fm * np.sin(2 * np.pi * fm * t) + ph)
    gate = (np.floor(t * 2 * 2) % 2 == 0).astype(float)
    x = x * gate + 0.001 * rng.randn(n) * gate
    pcm = np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    return pcm.astype(np.float32) / np.float32(32768)


def synth_clicks(pos, amp, tail_db=-50.0, seed=1234, sr=11025):
    """1 s of digital silence holding the samples amp[i] at pos[i], then 4 s of noise at tail_db dBFS, int16-quantised:
    the sparse-frame class (AFP_UNIT_TIE, include/afp.h)."""
    rng = np.random.RandomState(seed)
    x = np.concatenate([np.zeros(sr), rng.randn(4 * sr) * 10.0 ** (tail_db / 20.0)])
    for p_, a_ in zip(pos, amp):
        x[int(p_)] = a_
    pcm = np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    return pcm.astype(np.float32) / np.float32(32768)


def synth_fade(seed, level, secs_flat=2.0, secs_fade=3.0, secs_zero=1.0, sr=11025):
    """Noise at `level` (linear, of full scale), a LINEAR fade to zero, then digital silence -- quantised to int16 WITHOUT
    dither, so the last frames of the fade hold a few +-1 LSB samples (what a mastered fade-out looks like)."""
    rng = np.random.RandomState(seed)
    n0, n1, n2 = int(sr * secs_flat), int(sr * secs_fade), int(sr * secs_zero)
    env = np.concatenate([np.ones(n0), np.linspace(1.0, 0.0, n1), np.zeros(n2)])
    x = rng.randn(n0 + n1 + n2) * level * env
    pcm = np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
    return pcm.astype(np.float32) / np.float32(32768)


def sparse_parity_frames(d, n_fft=N_FFT, n_hop=N_HOP):
    """Frames of stft.stft's framing (stft.py:87-94) ALL of whose non-zero samples sit at offsets of one parity, with the
    level bound sum |x[q]| w[q] of each and the unit's floor max|S| / 1e6 (audfprint_analyze.py:285).  In exact arithmetic
    such a frame has |S(k)| == |S(256 - k)| (one sample: every bin equal), so the reference's own peak pick in it is
    rounding noise of its FFT.  Returns (frames above the floor, all single-parity frames, bound per frame, floor)."""
    d = np.asarray(d)
    if d.shape[0] == 0:
        return [], [], np.zeros(0), 0.0
    w = hann_window(n_fft)
    x = np.pad(d.astype(np.float64), n_fft // 2, mode='reflect')
    nfr = 1 + (x.shape[0] - n_fft) // n_hop
    idx = (np.arange(nfr)[:, None] * n_hop) + np.arange(n_fft)[None, :]
    fr = x[idx]
    nz = fr != 0
    ev = nz[:, 0::2].any(axis=1)
    od = nz[:, 1::2].any(axis=1)
    single = (ev ^ od)
    bound = (np.abs(fr) * w).sum(axis=1)
    smax = np.max(np.abs(stft_complex(d, n_fft, n_hop, w)))
    floor = smax / 1e6
    allf = np.flatnonzero(single).tolist()
    return [t for t in allf if bound[t] > floor], allf, bound, floor


# ---- "next" row f1: HashTable.store, hash_table.py:61-83, 91-138, 325-344 (test infrastructure) ----
class OracleHashTable(object):
    """The fields of the reference HashTable that store() touches, and store() itself restated."""

    def __init__(self, hashbits=20, depth=100, maxtime=16384):
        self.hashbits = hashbits
        self.depth = depth
        self.maxtimebits = int(round(np.log2(maxtime)))
        self.table = np.zeros((2 ** hashbits, depth), dtype=np.uint32)
        self.counts = np.zeros(2 ** hashbits, dtype=np.int32)
        self.names = []
        self.hashesperid = np.zeros(0, np.uint32)
        self.dirty = True

    def name_to_id(self, name, add_if_missing=False):                # hash_table.py:325-344
        if isinstance(name, str):
            if name not in self.names:
                if not add_if_missing:
                    raise ValueError("name " + name + " not found")
                try:
                    id_ = self.names.index(None)
                    self.names[id_] = name
                    self.hashesperid[id_] = 0
                except ValueError:
                    self.names.append(name)
                    self.hashesperid = np.append(self.hashesperid, [0])
            id_ = self.names.index(name)
        else:
            id_ = name
        return id_

    def store(self, name, timehashpairs, rng):                       # hash_table.py:91-138
        id_ = self.name_to_id(name, add_if_missing=True)
        hashmask = (1 << self.hashbits) - 1
        timemask = (1 << self.maxtimebits) - 1
        idval = (id_ + 1) << self.maxtimebits
        for time_, hash_ in timehashpairs:
            hash_ = int(hash_) & hashmask
            count = int(self.counts[hash_])
            val = idval + (int(time_) & timemask)
            if count < self.depth:
                self.table[hash_, count] = val
            else:
                slot = rng.randint(0, count)
                if slot < self.depth:
                    self.table[hash_, slot] = val
            self.counts[hash_] = count + 1
        self.hashesperid[id_] += len(timehashpairs)
        self.dirty = True

    def store_fast(self, name, timehashpairs, rng):                  # hash_table.py:91-138, the loop of store() batched per call
        """store() for long jobs (bench.py checks a 12 500-clip table: 8 M rows of the loop above would take half a minute).
        Same table, same counts, same draws in the same order: a row's count at insertion time is counts[hash] + its rank
        among the call's rows of that hash; rows that meet room (:121-124) write distinct slots, so they are written at once;
        rows that meet a full bucket (:125-131) always FOLLOW the fitting rows of their bucket and are replayed one by one in
        row order with the same rng.randint(0, count).  tests/test_table_build.py holds it against store() row for row."""
        id_ = self.name_to_id(name, add_if_missing=True)
        thp = np.asarray(timehashpairs).reshape(-1, 2)
        n = len(thp)
        if n:
            hashmask = (1 << self.hashbits) - 1
            timemask = (1 << self.maxtimebits) - 1
            idval = (id_ + 1) << self.maxtimebits
            h = thp[:, 1].astype(np.int64) & hashmask
            val = (idval + (thp[:, 0].astype(np.int64) & timemask)).astype(np.uint32)
            order = np.argsort(h, kind='stable')
            hs = h[order]
            first = np.r_[True, hs[1:] != hs[:-1]]
            idx = np.arange(n)
            rank = np.empty(n, np.int64)
            rank[order] = idx - np.maximum.accumulate(np.where(first, idx, 0))
            count = self.counts[h].astype(np.int64) + rank
            fit = count < self.depth
            self.table[h[fit], count[fit]] = val[fit]
            for i in np.nonzero(~fit)[0].tolist():
                slot = rng.randint(0, int(count[i]))
                if slot < self.depth:
                    self.table[h[i], slot] = val[i]
            starts = np.nonzero(first)[0]
            self.counts[hs[starts]] += np.diff(np.r_[starts, n]).astype(self.counts.dtype)
        self.hashesperid[id_] += n
        self.dirty = True

    def merge(self, ht, nprng):                                      # hash_table.py:291-323
        """nprng: a numpy RandomState standing in for the global np.random the reference draws from (:312)."""
        assert self.maxtimebits == ht.maxtimebits
        ncurrent = len(self.names)
        self.names += ht.names
        self.hashesperid = np.append(self.hashesperid, ht.hashesperid)
        idoffset = (1 << self.maxtimebits) * ncurrent
        for hash_ in np.nonzero(ht.counts)[0]:
            allvals = np.r_[self.table[hash_, :self.counts[hash_]],
                            ht.table[hash_, :ht.counts[hash_]] + idoffset]
            if len(allvals) > self.depth:
                somevals = nprng.permutation(allvals)[:self.depth]
                self.table[hash_, ] = somevals
                self.counts[hash_] += ht.counts[hash_]
            else:
                self.table[hash_, :len(allvals)] = allvals
                self.counts[hash_] = len(allvals)
        self.dirty = True

    def get_hits(self, hashes):                                      # hash_table.py:150-176
        nhashes = np.shape(hashes)[0]
        hits = np.zeros((nhashes * self.depth, 4), np.int32)
        nhits = 0
        maxtimemask = (1 << self.maxtimebits) - 1
        hashmask = (1 << self.hashbits) - 1
        for ix in range(nhashes):
            time_ = hashes[ix][0]
            hash_ = hashmask & hashes[ix][1]
            nids = min(self.depth, self.counts[hash_])
            tabvals = self.table[hash_, :nids]
            hitrows = nhits + np.arange(nids)
            hits[hitrows, 0] = (tabvals >> self.maxtimebits) - 1
            hits[hitrows, 1] = (tabvals & maxtimemask) - time_
            hits[hitrows, 2] = hash_
            hits[hitrows, 3] = time_
            nhits += nids
        return hits[:nhits].copy()


# ---- "next" row f4, second half: the vote counting of the matcher (test infrastructure) -------------
# audfprint_match.py:124-147 (_best_count_ids), :241-312 (_approx_match_counts, find_time_range off),
# :314-352 (match_hashes, exact_count off), helpers locmax :51-67 and keep_local_maxes :70-75.
def match_unique_hashes(hits, id_, mode, window=1):                     # audfprint_match.py:149-171
    """Matcher._unique_match_hashes: the unique (orig_time, hash) rows of the hits of `id_` within `window` of `mode`."""
    allids, alltimes = hits[:, 0], hits[:, 1]
    allhashes = hits[:, 2].astype(np.int64)
    allotimes = hits[:, 3]
    timebits = max(1, int(np.ceil(np.log(np.amax(allotimes)) / np.log(2))))      # encpowerof2 (:45-47)
    matchix = np.nonzero(np.logical_and(allids == id_, np.less_equal(np.abs(alltimes - mode), window)))[0]
    key = np.unique(allotimes[matchix] + (allhashes[matchix] << timebits))
    return np.c_[key & ((1 << timebits) - 1), key >> timebits]


def match_best_count_ids(hits, hashesperid, threshcount=5, search_depth=100):
    allids = hits[:, 0]
    ids = np.unique(allids)
    rawcounts = np.bincount(allids)[ids]                                  # :132
    weighted = rawcounts / (np.asarray(hashesperid)[ids].astype(float))  # :136
    order = np.argsort(weighted)[::-1]                                    # :139
    depth = np.minimum(np.count_nonzero(np.greater(rawcounts, threshcount)), search_depth)
    order = order[:depth]
    return ids[order], rawcounts[order]


def _match_keep_local_maxes(vec):                                         # :70-75 with locmax :51-67
    nbr = np.zeros(len(vec) + 1, dtype=bool)
    nbr[0] = True
    nbr[1:-1] = np.greater_equal(vec[1:], vec[:-1])
    ix = np.nonzero(nbr[:-1] & ~nbr[1:])[0]
    out = np.zeros(vec.shape)
    out[ix] = vec[ix]
    return out


def match_approx_counts(hits, ids, rawcounts, window=1, threshcount=5, max_alignments_per_id=100):
    results = np.zeros((len(ids), 7), np.int32)
    if not hits.size:
        return results
    allids = hits[:, 0].astype(int)
    alltimes = hits[:, 1].astype(int)
    mintime = np.amin(alltimes)                                           # :281
    alltimes = alltimes - mintime
    n = 0
    for urank, (id_, rawcount) in enumerate(zip(ids, rawcounts)):
        id_ = int(id_)
        bincounts = np.bincount(alltimes[allids == id_])                  # :289
        kept = _match_keep_local_maxes(bincounts)
        found = 0
        while True:
            mode = np.argmax(kept)
            if kept[mode] <= threshcount:
                break
            lo, hi = max(0, mode - window), mode + window + 1
            results[n, :] = [id_, np.sum(bincounts[lo:hi]), mode + mintime, rawcount, urank, 0, 0]
            n += 1
            if n >= results.shape[0]:
                results = np.vstack([results, np.zeros(results.shape, np.int32)])
            kept[lo:hi] = 0
            found += 1
            if found > max_alignments_per_id:
                break
    return results[:n, :]


def match_find_modes(data, threshold=5, window=0):                        # audfprint_match.py:78-90 (window is ignored there too)
    datamin = np.amin(data)
    fullvector = np.bincount(data - datamin)
    nbr = np.zeros(len(fullvector) + 1, dtype=bool)                       # locmax, :51-67
    nbr[0] = True
    nbr[1:-1] = np.greater_equal(fullvector[1:], fullvector[:-1])
    localmaxes = np.nonzero((nbr[:-1] & ~nbr[1:]) & np.greater_equal(fullvector, threshold))[0]
    return localmaxes + datamin, fullvector[localmaxes]


def match_unique_hashes(hits, id_, mode, window):                         # :149-171
    allids, alltimes, allhashes, allotimes = hits[:, 0], hits[:, 1], hits[:, 2].astype(np.int64), hits[:, 3]
    timebits = max(1, int(np.ceil(np.log(max(1, int(np.amax(allotimes)))) / np.log(2))))      # encpowerof2, :45-47, :157
    matchix = np.nonzero(np.logical_and(allids == id_, np.less_equal(np.abs(alltimes - mode), window)))[0]
    packed = np.unique(allotimes[matchix] + (allhashes[matchix] << timebits))
    return np.c_[packed & ((1 << timebits) - 1), packed >> timebits]


def match_time_range(sorted_hits, id_, mode, window, time_quantile=0.02):     # :173-193
    sel = np.logical_and.reduce([sorted_hits[:, 1] >= mode - window, sorted_hits[:, 1] <= mode + window, sorted_hits[:, 0] == id_])
    match_times = sorted_hits[sel, 3]
    return (match_times[int(len(match_times) * time_quantile)], match_times[int(len(match_times) * (1.0 - time_quantile)) - 1])


def match_exact_counts(hits, ids, rawcounts, window=1, threshcount=5, find_time_range=False, time_quantile=0.02):   # :195-239
    sorted_hits = hits[np.argsort(hits[:, 3], kind='stable')]             # (only the ORDER of equal times differs from :208; nothing read depends on it)
    allids, alltimes = sorted_hits[:, 0], sorted_hits[:, 1]
    results = np.zeros((max(1, len(ids) * 4), 7), np.int32)
    n = 0
    min_time = max_time = 0
    for urank, (id_, rawcount) in enumerate(zip(ids, rawcounts)):
        modes, _ = match_find_modes(alltimes[np.nonzero(allids == id_)[0]], window=window, threshold=threshcount)
        for mode in modes:
            filtcount = len(match_unique_hashes(sorted_hits, id_, mode, window))
            if filtcount >= threshcount:
                if n == results.shape[0]:
                    results = np.vstack([results, np.zeros(results.shape, np.int32)])
                if find_time_range:
                    min_time, max_time = match_time_range(sorted_hits, id_, mode, window, time_quantile)
                results[n, :] = [id_, filtcount, mode, rawcount, urank, min_time, max_time]
                n += 1
    return results[:n, :]


def match_hashes(ht, hashes, window=1, threshcount=5, search_depth=100, max_alignments_per_id=100, exact_count=False,
                 find_time_range=False, time_quantile=0.02):
    """Matcher.match_hashes (:314-352): the default switches, exact_count (:195-239) and find_time_range (:173-193, :300-302)."""
    hits = ht.get_hits(hashes)
    ids, raw = match_best_count_ids(hits, ht.hashesperid, threshcount, search_depth)
    if exact_count:
        res = match_exact_counts(hits, ids, raw, window, threshcount, find_time_range, time_quantile)
    else:
        res = match_approx_counts(hits, ids, raw, window, threshcount, max_alignments_per_id)
        if find_time_range and len(res):
            sorted_hits = hits[np.argsort(hits[:, 3], kind='stable')]
            for r in range(len(res)):
                res[r, 5:7] = match_time_range(sorted_hits, res[r, 0], res[r, 2], window, time_quantile)
    return res[(-res[:, 1]).argsort(), ]                                  # :336
