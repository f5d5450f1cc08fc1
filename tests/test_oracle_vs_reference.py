"""CPU, build container only: the oracle against the LIVE reference imported from
/root/reference (skipped where the tree is absent, e.g. on the GPU box)."""
import os
import sys

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')

from oracle import afp_oracle as O  # noqa: E402


@pytest.fixture(scope='module')
def ref():
    sys.path.insert(0, REF)
    try:
        import audfprint_analyze as A
        yield A
    finally:
        sys.path.remove(REF)
        for m in ('audfprint_analyze', 'stft', 'audio_read', 'hash_table'):
            sys.modules.pop(m, None)


def _ref_extract(A, d, prm):
    an = A.Analyzer(prm.density)
    an.maxpksperframe, an.maxpairsperpeak, an.f_sd, an.shifts = \
        prm.maxpksperframe, prm.maxpairsperpeak, prm.f_sd, prm.shifts
    pls = [an.find_peaks(d[o:], 11025) for o in O.shift_offsets(prm.shifts)]
    hs = np.concatenate([A.landmarks2hashes(an.peaks2landmarks(p)) for p in pls])
    return pls, (O.unique_sort_hashes(hs) if len(hs) else np.zeros((0, 2), np.int32))


@pytest.mark.parametrize('seed', range(6))
def test_random_configs(ref, seed):
    rng = np.random.RandomState(1000 + seed)
    prm = O.Params(density=float(rng.choice([10, 20, 70, 150])),
                   maxpksperframe=int(rng.choice([1, 3, 5, 9])),
                   maxpairsperpeak=int(rng.choice([1, 3, 10])),
                   f_sd=float(rng.choice([10.0, 30.0, 50.0])),
                   shifts=int(rng.choice([1, 1, 2, 4])))
    secs = float(rng.uniform(0.5, 6))
    d = O.synth_noise(seed, secs) if seed % 2 else O.synth_tonal(seed, secs)
    rp, rh = _ref_extract(ref, d, prm)
    op, oh = O.extract(d, prm)
    for a, b in zip(rp, op):
        assert np.array_equal(np.array(a, dtype=np.int32).reshape(-1, 2), b)
    assert np.array_equal(rh, oh)


def test_file_format_constants(ref):
    assert ref.PRECOMPEXT == '.afpt' and ref.PRECOMPPKEXT == '.afpk'
    assert ref.HASH_MAGIC == b'audfprinthashV00' and ref.PEAK_MAGIC == b'audfprintpeakV00'


@pytest.mark.parametrize('seed', range(64))
def test_the_gpu_suites_random_cases_hold_for_the_oracle_against_the_live_reference(ref, seed):
    """tests/test_gpu_random_configs.py compares the GPU with the ORACLE on 64 drawn (parameters, clips) cases; here the very same cases hold
    the oracle against the LIVE reference (targetdf / mindt / targetdt are Analyzer
    attributes there too, audfprint_analyze.py:139-143)."""
    from test_gpu_random_configs import draw
    kw, clips = draw(seed)
    prm = O.Params(**kw)
    an = ref.Analyzer(prm.density)
    an.maxpksperframe, an.maxpairsperpeak, an.f_sd, an.shifts = prm.maxpksperframe, prm.maxpairsperpeak, prm.f_sd, prm.shifts
    an.targetdf, an.mindt, an.targetdt = prm.targetdf, prm.mindt, prm.targetdt
    for d in clips:
        if len(d) > 70000:
            d = d[:70000]                                   # (keeps the CPU suite short; the GPU test runs the full length)
        pls, hs = O.extract(d, prm)
        rp = [an.find_peaks(d[o:], 11025) for o in O.shift_offsets(prm.shifts)]
        for a, b in zip(rp, pls):
            assert np.array_equal(np.array(a, dtype=np.int32).reshape(-1, 2), b)
        lms = [ref.landmarks2hashes(an.peaks2landmarks(p)) for p in rp]
        allh = np.concatenate(lms) if lms else np.zeros((0, 2), np.int32)
        want = O.unique_sort_hashes(allh) if len(allh) else np.zeros((0, 2), np.int32)
        assert np.array_equal(want, hs)


def test_unusual_sample_values_oracle_equals_the_live_reference(ref):
    """The clips of tests/test_gpu_corners.py::test_unusual_sample_values_equal_the_oracle (tiny / huge gain, DC offset,
    float32 denormals, full-scale square wave, a huge spike in quiet noise): the oracle against the LIVE reference."""
    base = O.synth_noise(4242, 6.0)
    n = len(base)
    spike = (base * np.float32(1e-3)).copy()
    spike[n // 2] = 0.9
    den = (base * np.float32(1e-6)).copy()
    den[100:4000] = np.float32(1e-40) * np.sign(base[100:4000])
    sq = np.where((np.arange(n) // 37) % 2 == 0, 1.0, -1.0).astype(np.float32)
    clips = [base * np.float32(1e-6), base * np.float32(1e4), (base * np.float32(0.3) + np.float32(0.5)).astype(np.float32),
             den, sq, spike]
    prm = O.Params()
    for i, d in enumerate(clips):
        rp, rh = _ref_extract(ref, d, prm)
        op, oh = O.extract(d, prm)
        assert np.array_equal(np.array(rp[0], dtype=np.int32).reshape(-1, 2), op[0]), i
        assert np.array_equal(rh, oh), i
