"""CPU: the FFT data flow of k_stft (lane/register index algebra, twiddle exponents, LDS
addressing, radix-8 butterflies, two-real-frames split) emulated on the host with the SAME
header the kernel compiles (audfprint_amd/csrc/fft512_core.h) and checked against numpy;
plus a bank-conflict audit of the LDS exchange layouts under the gfx950 ds_read_b64 / ds_write_b64 lane groups
(MI355X_MICROARCH.md §LDS)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'emul', 'fft_emul.cpp')
OUT = os.path.join(ROOT, 'tests', 'emul', '_build')


@pytest.fixture(scope='module')
def emul():
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, 'libfft_emul.so')
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-o', so, SRC])
    return ctypes.CDLL(so)


@pytest.mark.parametrize('fn', ['emul_stft_pair', 'emul_stft_pair_gen'])
@pytest.mark.parametrize('seed', range(4))
def test_two_real_frames_per_complex_fft(emul, seed, fn):
    """fn = ..._gen forms the twiddles as the kernel does (generators + repeated products); the other takes
    every twiddle from the table.  Both stay inside 1e-13 of numpy's power spectrum."""
    rng = np.random.RandomState(seed)
    xa, xb = rng.randn(512), rng.randn(512) * (10.0 ** rng.uniform(-3, 3))
    if seed == 3:
        xb[:] = 0.0
    pa, pb = np.zeros(257), np.zeros(257)
    P = ctypes.POINTER(ctypes.c_double)
    getattr(emul, fn)(xa.ctypes.data_as(P), xb.ctypes.data_as(P), pa.ctypes.data_as(P), pb.ctypes.data_as(P))
    ra, rb = np.abs(np.fft.rfft(xa)) ** 2, np.abs(np.fft.rfft(xb)) ** 2
    scale = max(ra.max(), rb.max())
    assert np.max(np.abs(pa - ra)) <= 1e-13 * scale
    assert np.max(np.abs(pb - rb)) <= 1e-13 * scale


def _strides():
    h = open(os.path.join(ROOT, 'audfprint_amd', 'csrc', 'fft512_core.h')).read()
    return int(re.search(r'#define FFT_X1_STRIDE (\d+)', h).group(1)), int(re.search(r'#define FFT_X2_STRIDE (\d+)', h).group(1))


# lane groups the LDS serves in one cycle (MI355X_MICROARCH.md, LDS table): ds_read_b64 2 x 32 lanes over 64 banks
# (32 slots of 8 bytes), ds_write_b64 4 x 16 contiguous lanes over 32 banks (16 slots of 8 bytes)
READ_GROUPS = [list(range(0, 32)), list(range(32, 64))]
WRITE_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]


def _max_conflict(addr_of_lane, groups, nslots):
    worst = 1
    for grp in groups:
        slots = {}
        for lane in grp:
            slots.setdefault(addr_of_lane(lane) % nslots, set()).add(addr_of_lane(lane))
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def test_lds_exchange_layouts_are_conflict_free():
    s1, s2 = _strides()
    for r in range(8):     # register index a / j / b
        assert _max_conflict(lambda L: r * s1 + L, WRITE_GROUPS, 16) == 1                      # xchg 1 write
        assert _max_conflict(lambda L: (L >> 3) * s1 + 8 * r + (L & 7), READ_GROUPS, 32) == 1  # xchg 1 read
        assert _max_conflict(lambda L: (L & 7) * s2 + 8 * r + (L >> 3), WRITE_GROUPS, 16) == 1  # xchg 2 write
        assert _max_conflict(lambda L: r * s2 + L, READ_GROUPS, 32) == 1                       # xchg 2 read
    # both exchanges and the 16 parked Nyquist values fit the per-wavefront buffer
    h = open(os.path.join(ROOT, 'audfprint_amd', 'csrc', 'fft512_core.h')).read()
    assert '#define FFT_LDS_DOUBLES (8 * FFT_X1_STRIDE + 16)' in h
    assert 7 * s1 + 63 < 8 * s1 and 7 * s2 + 63 < 8 * s1
