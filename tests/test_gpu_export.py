"""GPU: small batches (one file per call, audfprint.py:164-165) end with k_export -- rows, offsets, unit flags and totals
written into pinned host memory by one launch, afp_fetch_all = one wait + a host memcpy.  Checked against a handle with
the export switched off (AFP_EXPORT_MAX_UNITS=0: the five device-to-host copies), against the goldens, on the first call
of a fresh handle (output estimate too small: the scatter is re-run and the image must be ignored), and on a result that
does not fit the image."""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _fresh(max_units=None):
    """A NEW library handle (not the per-process singleton): the export limit is read when the handle is created."""
    from audfprint_amd.batch import Extractor
    old = os.environ.get('AFP_EXPORT_MAX_UNITS')
    try:
        if max_units is None:
            os.environ.pop('AFP_EXPORT_MAX_UNITS', None)
        else:
            os.environ['AFP_EXPORT_MAX_UNITS'] = str(max_units)
        return Extractor(0)
    finally:
        if old is None:
            os.environ.pop('AFP_EXPORT_MAX_UNITS', None)
        else:
            os.environ['AFP_EXPORT_MAX_UNITS'] = old


def _same(a, b, nclips, shifts):
    for f in ('hashes', 'hash_offsets', 'peaks', 'peak_offsets', 'unit_flags'):
        x, y = getattr(a, f), getattr(b, f)
        assert (x is None) == (y is None), f
        if x is not None:
            assert np.array_equal(x, y), f


def test_export_image_equals_the_copied_results():
    from oracle import afp_oracle as O
    on, off = _fresh(), _fresh(0)
    try:
        rng = np.random.RandomState(5)
        for shifts, nclips in ((1, 1), (1, 7), (4, 3), (2, 32), (1, 64), (1, 65), (4, 17)):
            for e in (on, off):
                e.set_params(shifts=shifts)
            clips = [O.synth_noise(300 + i, 1.0 + 7.0 * rng.rand()) for i in range(nclips)]
            if nclips > 2:
                clips[1] = np.zeros(0, np.float32)                      # an empty clip inside the batch
                clips[2] = np.zeros(5000, np.float32)                   # digital silence: UNIT_ZERO
            for wh, wp in ((True, True), (True, False), (False, True)):
                a = on.extract(clips=clips, want_hashes=wh, want_peaks=wp)
                b = off.extract(clips=clips, want_hashes=wh, want_peaks=wp)
                _same(a, b, nclips, shifts)
                if wh:
                    assert a.hash_offsets[-1] == len(a.hashes) and (nclips == 1 or len(a.hashes) > 0)
    finally:
        on.close()
        off.close()


@pytest.mark.parametrize('name', ['noise_s0_10s', 'noise_s0_30s_c5', 'tonal_s4_12s_c5', 'hand_silence_then_noise', 'noise_n257'])
def test_first_call_of_a_fresh_handle_and_the_steady_state(name):
    """First call: the output buffers are sized from an estimate; with the C5 parameters it is too small, finalize() re-runs
    the scatter and the pinned image (taken before that) must not be used.  Second call: buffers fit, the image is used."""
    g = load_golden(name)
    e = _fresh()
    try:
        p = g['params']
        e.set_params(density=p['density'], maxpksperframe=p['maxpksperframe'], maxpairsperpeak=p['maxpairsperpeak'],
                     f_sd=p['f_sd'], shifts=p['shifts'], targetdf=p['targetdf'], mindt=p['mindt'], targetdt=p['targetdt'])
        for call in range(3):
            r = e.extract(clips=[g['d']], want_hashes=True, want_peaks=True)
            assert np.array_equal(r.clip_hashes(0), g['hashes']), (name, call)
            for s in range(max(1, p['shifts'])):
                assert np.array_equal(r.unit_peaks(0, s), g['peaks'][s]), (name, call, s)
    finally:
        e.close()


def test_a_result_larger_than_the_image_takes_the_copies():
    """300 s at the C5 parameters: ~0.6 M hash rows = 4.7 MB, more than the 4 MB image -- k_export writes only the totals and
    afp_fetch_all falls back to the device-to-host copies."""
    from oracle import afp_oracle as O
    on, off = _fresh(), _fresh(0)
    try:
        d = O.synth_noise(9, 300.0)
        for e in (on, off):
            e.set_params(density=70.0, maxpairsperpeak=10, shifts=4)
        a = on.extract(clips=[d], want_hashes=True, want_peaks=True)
        a = on.extract(clips=[d], want_hashes=True, want_peaks=True)      # (second call: buffers already fit)
        b = off.extract(clips=[d], want_hashes=True, want_peaks=True)
        assert len(a.hashes) * 8 > (4 << 20)
        _same(a, b, 1, 4)
    finally:
        on.close()
        off.close()
