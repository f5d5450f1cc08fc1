"""GPU: BASELINE-size batches checked through size-independent properties (tiling/idempotence,
sortedness, uniqueness, determinism, CSR consistency) plus bit-exact oracle comparison on the
distinct clips of the pool."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ex():
    from audfprint_amd.batch import Extractor
    return Extractor.get(0)


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
    pool = [O.synth_noise(600 + i, 30.0) for i in range(4)]
    clips = [pool[i % 4] for i in range(256)]
    ex.set_params(**kw)
    r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    _check_csr_sorted_unique(r, len(clips))
    for i in range(4):
        pls, hs = O.extract(pool[i], O.Params(**kw))
        assert np.array_equal(r.clip_hashes(i), hs)
        for s in range(4):
            assert np.array_equal(r.unit_peaks(i, s), pls[s])
    for i in range(4, 256):
        assert np.array_equal(r.clip_hashes(i), r.clip_hashes(i % 4))


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
    with pytest.raises(_lib.AfpError):                      # columns must be non-decreasing
        ex.pairs_from_peaks([np.array([[5, 3], [2, 4]], np.int32)])
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
    finally:
        e2.close()
