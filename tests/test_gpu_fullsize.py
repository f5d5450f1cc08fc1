"""GPU: BASELINE-size batches checked through size-independent properties (tiling/idempotence,
sortedness, uniqueness, determinism, CSR consistency) plus bit-exact oracle comparison on the
distinct clips of the pool."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ex():
    from audfprint_amd.batch import Extractor
    return Extractor.get(0)


def _oracle_c5(seed):
    from oracle import afp_oracle as O
    return O.extract(O.synth_noise(seed, 30.0), O.Params(density=70.0, maxpairsperpeak=10, shifts=4))


def _keys(h):
    return (h[:, 0].astype(np.int64) << 32) | h[:, 1].astype(np.int64)


def _check_csr_sorted_unique(r, nclips):
    off = r.hash_offsets
    assert off[0] == 0 and off[-1] == len(r.hashes) and np.all(np.diff(off) >= 0)
    k = _keys(r.hashes)
    d = np.diff(k)
    starts = off[1:-1]
    inner = np.ones(len(k) - 1, dtype=bool) if len(k) > 1 else np.zeros(0, bool)
    inner[starts[(starts > 0) & (starts < len(k))] - 1] = False          # boundaries between clips
    assert np.all(d[inner] > 0), 'rows of a clip must be strictly increasing in (time, hash)'
    assert r.hashes[:, 1].min() >= 0 and r.hashes[:, 1].max() < (1 << 20) and r.hashes[:, 0].min() >= 0


def test_c3_size_batch_1024x30s(ex):
    from oracle import afp_oracle as O
    npool, nclips, secs = 16, 1024, 30.0
    pool = [O.synth_noise(500 + i, secs) for i in range(npool)]
    clips = [pool[i % npool] for i in range(nclips)]
    ex.set_params()
    r1 = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    _check_csr_sorted_unique(r1, nclips)
    for i in range(npool):
        pls, hs = O.extract(pool[i], O.Params())
        assert np.array_equal(r1.clip_hashes(i), hs) and np.array_equal(r1.unit_peaks(i), pls[0])
    # tiling: every copy of a pool clip gives the identical rows
    for i in range(npool, nclips):
        assert np.array_equal(r1.clip_hashes(i), r1.clip_hashes(i % npool)), i
    # determinism: checksum of checksums over two more runs
    digest = hashlib.sha256(r1.hashes.tobytes() + r1.hash_offsets.tobytes() + r1.peaks.tobytes()).hexdigest()
    for _ in range(2):
        r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        assert hashlib.sha256(r.hashes.tobytes() + r.hash_offsets.tobytes() + r.peaks.tobytes()).hexdigest() == digest
    # frame bound: times < T = 1 + N//256
    assert r1.hashes[:, 0].max() < 1 + len(pool[0]) // 256


def test_c5_parameters_256x30s(ex):
    from oracle import afp_oracle as O
    kw = dict(density=70.0, maxpairsperpeak=10, shifts=4)
    # 64 DISTINCT clips, every one against the oracle (run over the host's cores), tiled to 256
    from concurrent.futures import ProcessPoolExecutor
    npool = 64
    pool = [O.synth_noise(600 + i, 30.0) for i in range(npool)]
    clips = [pool[i % npool] for i in range(256)]
    ex.set_params(**kw)
    r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    _check_csr_sorted_unique(r, len(clips))
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as pe:
        want = list(pe.map(_oracle_c5, [600 + i for i in range(npool)]))
    for i in range(npool):
        pls, hs = want[i]
        assert np.array_equal(r.clip_hashes(i), hs), i
        for s in range(4):
            assert np.array_equal(r.unit_peaks(i, s), pls[s]), (i, s)
    for i in range(npool, 256):
        assert np.array_equal(r.clip_hashes(i), r.clip_hashes(i % npool))


def test_ragged_c4_like_batch(ex):
    """5000 clips of 0.02-12 s (some empty, some all-zero): a random sample against the oracle."""
    from oracle import afp_oracle as O
    rng = np.random.RandomState(7)
    base = [O.synth_noise(700 + i, 12.0) for i in range(8)]
    clips = []
    for i in range(5000):
        n = int(rng.choice([0, 200, 256, 2000, 11025, 60000, 110250, 132300], p=[.01, .02, .02, .05, .2, .3, .3, .1]))
        clips.append(np.zeros(n, np.float32) if i % 97 == 0 else base[i % 8][:n])
    ex.set_params()
    r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    _check_csr_sorted_unique(r, len(clips))
    for i in rng.choice(len(clips), 40, replace=False):
        pls, hs = O.extract(clips[i], O.Params())
        assert np.array_equal(r.clip_hashes(i), hs), i
        assert np.array_equal(r.unit_peaks(i), pls[0]), i


def test_error_codes_and_limits(ex):
    import ctypes as C
    from audfprint_amd import _lib
    from audfprint_amd.batch import Extractor
    with pytest.raises(ValueError):
        ex.set_params(maxpksperframe=65)
    with pytest.raises(ValueError):
        ex.set_params(shifts=17)
    with pytest.raises(_lib.AfpError):
        ex.set_params(maxpairsperpeak=0)
    ex.set_params()
    with pytest.raises(_lib.AfpError):                      # bin out of range
        ex.pairs_from_peaks([np.array([[1, 300]], np.int32)])
    with pytest.raises(ValueError):                         # the last row must hold the largest column (:321)
        ex.pairs_from_peaks([np.array([[5, 3], [2, 4]], np.int32)])
    # list-order columns may hold any number of rows (round 6; the reference's peaks_at[col] has no limit, :321-341): a bin
    # listed 700 times, then twice out of order -- landmarks and hashes equal the oracle's
    from oracle import afp_oracle as O
    long_col = np.array([[2, 9]] * 700 + [[2, 40], [2, 12], [4, 30], [4, 11], [5, 1], [9, 7]], np.int32)
    r_l, lms = ex.pairs_from_peaks([long_col], want_hashes=True, want_landmarks=True)
    lm = O.peaks2landmarks(long_col, O.Params())
    assert np.array_equal(lms[0], lm.astype(np.int32)) and len(lm) >= 700 * 3
    assert np.array_equal(r_l.clip_hashes(0), O.unique_sort_hashes(O.landmarks2hashes(lm)))
    # columns out of order but the last row is the largest: stable-sorted by column on the host, same pairs as sorted input
    pk = np.array([[0, 10], [4, 30], [2, 20], [2, 40], [6, 25], [9, 22]], np.int32)
    r_a, _ = ex.pairs_from_peaks([pk])
    r_b, _ = ex.pairs_from_peaks([pk[np.argsort(pk[:, 0], kind='stable')]])
    assert np.array_equal(r_a.hashes, r_b.hashes) and len(r_a.hashes) > 0
    e2 = Extractor(0)
    try:
        e2.set_params()
        assert e2.lib.afp_fetch_hashes(e2.h, None, None) == -5           # AFP_ERR_STATE: nothing extracted yet
        assert e2.lib.afp_set_workspace_limit(e2.h, 1 << 20) == 0
        with pytest.raises(_lib.AfpError, match='workspace'):
            e2.extract(clips=[np.zeros(11025 * 30, np.float32)])
        assert e2.lib.afp_set_workspace_limit(e2.h, 1 << 34) == 0
        r = e2.extract(clips=[np.zeros(0, np.float32)])
        assert len(r.hashes) == 0 and r.hash_offsets.tolist() == [0, 0]
        # a clip of 2^21 frames is refused before anything is read (32-bit row offsets inside a unit)
        with pytest.raises(ValueError):                         # offsets beyond the buffer never reach the library
            e2.extract(pcm=np.zeros(16, np.float32), offsets=np.array([0, 256 * (1 << 21)], np.int64))
        big = np.array([0, 256 * (1 << 21)], np.int64)
        assert e2.lib.afp_extract_host(e2.h, np.zeros(16, np.float32).ctypes.data_as(C.POINTER(C.c_float)),
                                       big.ctypes.data_as(C.POINTER(C.c_int64)), 1, 1) == -1      # AFP_ERR_ARG, buffer not read
    finally:
        e2.close()


def test_near_tie_sensitivity_sweep_2000_mixed_clips(ex):
    """how often does the integer output depend on floating-point noise?  2048 short clips of the
    classes where ties and near-ties live -- tonal + gated (digital-silence plateaus), hard-clipped, very quiet
    (amplitudes of a few LSB), DC steps, sparse clicks on a noise bed, pure tones at exact bin centres -- against
    the oracle, integer-exact.  Expected flips: 0; units the library itself flags AFP_UNIT_TIE are exempt (and counted)."""
    from oracle import afp_oracle as O
    from audfprint_amd import _lib
    sr = 11025
    clips = []
    for i in range(2048):
        rng = np.random.RandomState(31000 + i)
        n = int(rng.randint(1, 5) * sr + rng.randint(0, 600))
        kind = i % 8
        if kind == 0:
            d = O.synth_tonal(31000 + i, n / float(sr))[:n]
        elif kind == 1:                                              # hard-clipped noise
            d = np.clip(rng.randn(n) * 2.0, -1, 1)
        elif kind == 2:                                              # a few LSB of noise
            d = np.round(rng.randn(n) * 1.5) / 32768.0
        elif kind == 3:                                              # DC steps + tiny noise
            d = np.repeat(rng.randint(-8000, 8000, n // 2000 + 1), 2000)[:n] / 32768.0 + rng.randn(n) * 1e-4
        elif kind == 4:                                              # clicks on a quiet bed
            d = rng.randn(n) * 3e-4
            d[rng.randint(0, n, 12)] = rng.choice([-0.9, 0.9], 12)
        elif kind == 5:                                              # bin-centred tones, gated
            t = np.arange(n)
            d = sum(0.2 * np.sin(2 * np.pi * k * t / 512.0) for k in rng.choice(np.arange(4, 250), 3, replace=False))
            d = d * (np.floor(t / 1500.0) % 2)
        elif kind == 6:                                              # noise burst between silences
            d = np.zeros(n)
            a, b = sorted(rng.randint(0, n, 2))
            d[a:b] = rng.randn(b - a) * 0.05
        else:
            d = rng.randn(n) * 0.1
        pcm = np.round(np.clip(d, -1, 1) * 32767).astype(np.int16)
        clips.append(pcm.astype(np.float32) / np.float32(32768))
    flips, flagged, total_peaks = 0, 0, 0
    for kw in (dict(), dict(density=70.0, maxpairsperpeak=10)):
        ex.set_params(**kw)
        r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        prm = O.Params(**kw)
        for i, d in enumerate(clips):
            if r.unit_flags[i] & _lib.UNIT_TIE:
                flagged += 1
                continue
            pls, hs = O.extract(d, prm)
            total_peaks += len(pls[0])
            if not (np.array_equal(r.unit_peaks(i), pls[0]) and np.array_equal(r.clip_hashes(i), hs)):
                flips += 1
    print('near-tie sweep: %d peaks compared, %d clips differ, %d tie-flagged units' % (total_peaks, flips, flagged))
    assert flips == 0
    assert flagged <= 8            # lone clicks inside digital silence do not occur in these classes (the click class has a noise bed)
