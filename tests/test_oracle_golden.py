"""CPU: the oracle restatement must reproduce every golden fixture made from the live
reference (tests/golden/make_golden.py) -- integers bit-exact, float stages bit-exact too
(same numpy primitives in the same order)."""
import hashlib

import numpy as np
import pytest

from conftest import SPARSE_CONTROLS, SPARSE_FRAME, golden_names, load_golden
from oracle import afp_oracle as O

SLOW = {'noise_s0_300s'}


def _prm(p):
    return O.Params(**{k: p[k] for k in ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd',
                                          'shifts', 'targetdf', 'mindt', 'targetdt')})


@pytest.mark.parametrize('name', golden_names())
def test_oracle_matches_golden(name):
    g = load_golden(name)
    sha = hashlib.sha256(np.ascontiguousarray(g['d']).tobytes()).hexdigest()[:16]
    assert sha == g['pcm_sha'], 'synthetic input recipe drifted'
    prm = _prm(g['params'])
    peaklists, hashes = O.extract(g['d'], prm)
    assert len(peaklists) == len(g['peaks'])
    for a, b in zip(peaklists, g['peaks']):
        assert np.array_equal(a, b)
    assert hashes.dtype == np.int32 and np.array_equal(hashes, g['hashes'])
    lm = O.peaks2landmarks(peaklists[0], prm)
    assert np.array_equal(lm.astype(np.int32), g['landmarks0'])


@pytest.mark.parametrize('name', [n for n in golden_names() if n.endswith('_stages')
                                  or n == 'hand_silence_then_noise'])
def test_oracle_float_stages(name):
    g = load_golden(name)
    st = O.find_peaks_stages(g['d'], _prm(g['params']))
    assert np.array_equal(st['mag'], g['mag'])
    assert np.array_equal(st['logs'], g['logs'])
    assert np.array_equal(st['sgram'], g['sgram'])
    assert np.array_equal(np.packbits(st['fwd'].astype(bool), axis=0), g['fwd'])


@pytest.mark.parametrize('name', [n for n in golden_names() if n not in SLOW])
def test_sparse_frame_list_is_the_parity_rule(name):
    """conftest.SPARSE_FRAME (the fixtures a GPU test may exempt from bit-exactness when the library flags them) must be
    exactly the fixtures holding a frame whose non-zero samples share one parity and whose level rises above the floor
    (include/afp.h, AFP_UNIT_TIE) -- on every shift grid the fixture's parameters ask for."""
    g = load_golden(name)
    flagged = False
    for off in O.shift_offsets(g['params']['shifts']):
        above, _all, _bound, _floor = O.sparse_parity_frames(g['d'][off:])
        flagged = flagged or bool(above)
    assert flagged == (name in SPARSE_FRAME), (name, flagged)
    if name in SPARSE_CONTROLS:
        assert not flagged


def test_empty_and_zero_inputs():
    assert O.find_peaks(np.zeros(0, np.float32)).shape == (0, 2)
    pl, h = O.extract(np.zeros(11025, np.float32))
    assert pl[0].shape == (0, 2) and h.shape == (0, 2)


def test_kat_table_of_survey():
    """SURVEY.md §8c KAT rows (counts and first/last entries)."""
    pl, h = O.extract(O.synth_noise(0, 10))
    assert len(pl[0]) == 233 and len(h) == 669
    assert pl[0][:3].tolist() == [[8, 32], [11, 62], [12, 176]]
    assert h[:3].tolist() == [[8, 131150], [8, 132572], [8, 132995]] and h[-1].tolist() == [426, 218116]
