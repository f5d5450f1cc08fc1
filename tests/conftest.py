"""pytest configuration: registers the `gpu` marker and offers golden-fixture helpers."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')
    # the test process is a host application like any other: it configures the runtime once, before anything touches HIP
    import audfprint_amd
    audfprint_amd.configure_runtime()


def pytest_sessionstart(session):
    """Make sure libafp_hip.so exists and is not older than its sources (hipcc cross-compiles gfx950
    without a GPU, so this works on the CPU box and on the GPU box alike)."""
    try:
        from audfprint_amd import build
        build.build(force=False, verbose=False)
    except Exception as e:  # no hipcc: tests that need the library will say so themselves
        print('conftest: could not (re)build libafp_hip.so: %r' % (e,))


# Fixtures of the sparse-frame class (AFP_UNIT_TIE, include/afp.h): a frame all of whose non-zero samples sit at offsets of one
# parity has bins that are EQUAL in exact arithmetic (|S(k)| == |S(256 - k)|; a lone click: every bin), and which of them the
# reference picks is rounding noise of its own FFT (tools/sparse_frame_jitter.py, profiles/r05_sparse_frame_jitter_reference.json).
# These are the ONLY fixtures a GPU test may treat differently from "bit-exact", and only when the library flags them;
# tests/test_oracle_golden.py checks that this list IS the rule (oracle.sparse_parity_frames) applied to every fixture.
LONE_CLICK = ('hand_impulse', 'hand_click_then_noise', 'hand_click_then_quiet_noise')
SPARSE_CLICKS = tuple('sparse_%s_%ddb' % (nm, db) for db in (50, 20) for nm in
                      ['two_d%d_%s' % (dl, an) for dl in (64, 100, 128, 256) for an in ('eq', 'uneq')] + ['three_s128', 'four_s64'])
SPARSE_FRAME = LONE_CLICK + SPARSE_CLICKS + ('fade_quiet',)
# controls: sparse frames holding BOTH parities (two clicks 101 apart, four clicks 63 apart) and a loud fade-out whose last
# frames are dense -- these must stay unflagged and bit-exact
SPARSE_CONTROLS = tuple('sparse_%s_%ddb' % (nm, db) for db in (50, 20) for nm in ('two_d101_eq', 'four_s63')) + ('fade_loud',)


def golden_names():
    with open(os.path.join(GOLDEN, 'INDEX.json')) as f:
        return sorted(json.load(f).keys())


def load_golden(name):
    """Returns dict(d=float32 PCM, params=dict, peaks=[(P,2) per shift], hashes=(N,2),
    landmarks0, plus optional float stages)."""
    from oracle import afp_oracle as O      # tests may use the oracle's input recipes
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(str(z['meta']))
    spec = meta['spec']
    if spec['kind'] == 'hand':
        d = z['pcm_i16'].astype(np.float32) / np.float32(32768)
    elif spec['kind'] == 'noise':
        d = O.synth_noise(spec['seed'], spec['secs'], nsamp=spec.get('nsamp'))
    elif spec['kind'] == 'clicks':
        d = O.synth_clicks(spec['pos'], spec['amp'], spec['tail_db'])
    elif spec['kind'] == 'fade':
        d = O.synth_fade(spec['seed'], spec['level'])
    else:
        d = O.synth_tonal(spec['seed'], spec['secs'])
    assert len(d) == meta['nsamp']
    out = dict(d=d, params=meta['params'], hashes=z['hashes'], landmarks0=z['landmarks0'],
               peaks=[z['peaks%d' % s] for s in range(max(1, meta['params']['shifts']))],
               pcm_sha=meta['pcm_sha'])
    for k in ('mag', 'logs', 'sgram', 'fwd'):
        if k in z.files:
            out[k] = z[k]
    return out


@pytest.fixture(scope='session')
def golden_loader():
    return load_golden
