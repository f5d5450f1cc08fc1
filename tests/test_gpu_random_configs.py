"""GPU: randomly drawn Analyzer configurations (the whole parameter space the C ABI accepts) and
signal types, full path vs the oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ex():
    from audfprint_amd.batch import Extractor
    e = Extractor.get(0)
    yield e
    e.set_pipeline()


# the kernel path is forced per case: by default these small batches would all take the same one.  'segments_short': a
# deliberately short warm-up -- segments are re-run by the repair launch or the sequential fallback takes over.
PATHS = {'dense': dict(compact=0, seg=0), 'compact': dict(compact=1, seg=0), 'segments': dict(compact=0, seg=1, seg_len=32),
         'segments_short': dict(compact=0, seg=1, seg_len=16, seg_warm=6)}


def _signal(rng, kind, n):
    from oracle import afp_oracle as O
    if kind == 'noise':
        return O.synth_noise(int(rng.randint(1 << 30)), 0, nsamp=n)
    if kind == 'tonal':
        return O.synth_tonal(int(rng.randint(1 << 30)), n / 11025.0)[:n]
    if kind == 'burst':          # noise bursts separated by digital silence
        x = O.synth_noise(int(rng.randint(1 << 30)), 0, nsamp=n).copy()
        for _ in range(4):
            a = rng.randint(0, max(1, n - 1))
            x[a:a + rng.randint(100, 9000)] = 0.0
        return x
    if kind == 'quiet':          # a few LSBs of amplitude: heavy flooring
        return (np.round(rng.randn(n) * 2.0).astype(np.int16).astype(np.float32)) / np.float32(32768)
    raise ValueError(kind)


def draw(seed):
    """(parameters, clips) of case `seed` -- shared with tests/test_oracle_vs_reference.py, which holds the ORACLE against the
    live reference on the very same cases (so reference -> oracle -> GPU is one chain per case)."""
    rng = np.random.RandomState(4000 + seed)
    kw = dict(density=float(rng.choice([5, 20, 35, 70, 150, 400])),
              maxpksperframe=int(rng.choice([1, 2, 5, 5, 9, 17, 64])),
              maxpairsperpeak=int(rng.choice([1, 3, 3, 8, 20])),
              f_sd=float(rng.choice([4.0, 15.0, 30.0, 30.0, 60.0, 200.0])),
              shifts=int(rng.choice([1, 1, 2, 4, 5, 16])),
              targetdf=int(rng.choice([1, 8, 31, 31, 33, 64])),
              mindt=int(rng.choice([0, 1, 2, 2, 5])),
              targetdt=int(rng.choice([3, 32, 63, 63, 64, 200])))
    clips = []
    for _ in range(int(rng.randint(1, 5))):
        n = int(rng.choice([0, 1, 255, 256, 700, 4000, 11025, 30000, 66150, 132300]))
        clips.append(_signal(rng, str(rng.choice(['noise', 'tonal', 'burst', 'quiet'])), n) if n else np.zeros(0, np.float32))
    return kw, clips


NCASES = 64


@pytest.mark.parametrize('seed', range(NCASES))
def test_random_configuration(ex, seed):
    from oracle import afp_oracle as O
    ex.set_pipeline(**PATHS[sorted(PATHS)[seed % len(PATHS)]])
    kw, clips = draw(seed)
    prm = O.Params(**kw)
    ex.set_params(**kw)
    r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    for i, d in enumerate(clips):
        pls, hs = O.extract(d, prm)
        for s in range(prm.shifts if prm.shifts >= 2 else 1):
            assert np.array_equal(r.unit_peaks(i, s), pls[s]), (kw, i, s, len(d))
        assert np.array_equal(r.clip_hashes(i), hs), (kw, i, len(d))
