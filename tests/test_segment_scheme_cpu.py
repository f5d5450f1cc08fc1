"""CPU: the premise of the segment-parallel scan (k_scan_seg, afp_common.h SegDesc), checked with the numpy oracle alone --
a forward / backward threshold pass that starts W frames away from a segment, from the STANDARD initialisation on its own
first column(s), holds at the segment's first frame a threshold vector BIT-identical to the sequential pass's (the
threshold is an element-wise max of decayed bumps: once newer bumps dominate every bin the old state is gone, and equal
states stay equal).  W is what the library uses: ceil(1 / (1 - a_dec)), at least 64."""
import math

import numpy as np
import pytest

from oracle import afp_oracle as O


def _fwd_state_at(sgram, a_dec, G, K, t0, t_query):
    """threshold at ENTRY of frame t_query of a forward pass started at frame t0 (audfprint_analyze.py:199-231)."""
    srows, scols = sgram.shape
    st = O.spreadpeaksinvector(np.max(sgram[:, t0:min(t0 + 10, scols)], axis=1), G)
    for col in range(t0, t_query):
        s_col = sgram[:, col]
        pos = np.nonzero(O.locmax(s_col) & (s_col > st))[0]
        for val, p in sorted(zip(s_col[pos], pos), reverse=True)[:K]:
            st = np.maximum(st, val * G[srows - p: 2 * srows - p])
        st = st * a_dec
    return st


def _bwd_state_at(sgram, fwd, a_dec, G, t_end, t_query):
    """threshold at ENTRY of frame t_query of a backward pass started at frame t_end - 1 (:233-253)."""
    srows = sgram.shape[0]
    st = O.spreadpeaksinvector(sgram[:, t_end - 1], G)
    for col in range(t_end, t_query + 1, -1):
        pk = np.nonzero(fwd[:, col - 1])[0]
        for val, p in sorted(zip(sgram[pk, col - 1], pk), reverse=True):
            if val >= st[p]:
                st = np.maximum(st, val * G[srows - p: 2 * srows - p])
        st = a_dec * st
    return st


@pytest.mark.parametrize('kind,density', [('noise', 20.0), ('tonal', 20.0), ('noise', 70.0), ('tonal', 70.0)])
def test_a_warm_up_of_one_decay_length_reaches_the_sequential_state(kind, density):
    d = O.synth_noise(91, 40.0) if kind == 'noise' else O.synth_tonal(92, 40.0)
    prm = O.Params(density=density)
    stg = O.find_peaks_stages(d, prm)
    sg, fwd = stg['sgram'], stg['fwd']
    G = O.gauss_table(256, prm.f_sd)
    a = O.a_dec_of(density)
    W = max(64, int(math.ceil(1.0 / (1.0 - a))))
    T = sg.shape[1]
    for s in range(W + 50, T - W - 50, 311):              # segment starts / ends scattered over the clip
        ref = _fwd_state_at(sg, a, G, prm.maxpksperframe, 0, s)
        got = _fwd_state_at(sg, a, G, prm.maxpksperframe, s - W, s)
        assert np.array_equal(ref, got), ('forward', kind, density, s)
        e = s
        ref = _bwd_state_at(sg, fwd, a, G, T, e)
        got = _bwd_state_at(sg, fwd, a, G, min(T, e + 1 + W), e)
        assert np.array_equal(ref, got), ('backward', kind, density, e)
