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


def pytest_sessionstart(session):
    """Make sure libafp_hip.so exists and is not older than its sources (hipcc cross-compiles gfx950
    without a GPU, so this works on the CPU box and on the GPU box alike)."""
    try:
        from audfprint_amd import build
        build.build(force=False, verbose=False)
    except Exception as e:  # no hipcc: tests that need the library will say so themselves
        print('conftest: could not (re)build libafp_hip.so: %r' % (e,))


# Fixtures of the lone-click class: a frame holding exactly ONE non-zero sample has a spectrum flat to the last bit, and which
# of its equal bins count as local maxima is FFT rounding noise in the reference (AFP_UNIT_TIE, include/afp.h).  These are the
# ONLY fixtures a GPU test may treat differently from "bit-exact", and only when the library flags them.
LONE_CLICK = ('hand_impulse', 'hand_click_then_noise', 'hand_click_then_quiet_noise')


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
