#!/usr/bin/env python
"""CPU experiment (numpy oracle): how many warm-up frames does a forward / backward threshold scan that starts in the
middle of a clip (standard initialisation on its own first columns) need before its state is BIT-identical to the
sequential scan's?  Decides the warm-up length of the segment-parallel scan (k_scan segments)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import afp_oracle as O


def fwd_states(sgram, a_dec, G, K, t0):
    srows, scols = sgram.shape
    st = O.spreadpeaksinvector(np.max(sgram[:, t0:min(t0 + 10, scols)], axis=1), G)
    out = []
    for col in range(t0, scols):
        out.append(st.copy())
        s_col = sgram[:, col]
        pos = np.nonzero(O.locmax(s_col) & (s_col > st))[0]
        for val, p in sorted(zip(s_col[pos], pos), reverse=True)[:K]:
            st = np.maximum(st, val * G[srows - p: 2 * srows - p])
        st = st * a_dec
    return out


def conv_len(sgram, a_dec, G, K, starts, maxw):
    ref = fwd_states(sgram, a_dec, G, K, 0)
    res = []
    for t0 in starts:
        seg = fwd_states(sgram[:, :min(sgram.shape[1], t0 + maxw)], a_dec, G, K, t0)
        w = None
        for i, s in enumerate(seg):
            if np.array_equal(s, ref[t0 + i]):
                w = i
                break
        res.append(w)
    return res


for name, d, dens in (('noise d20', O.synth_noise(3, 120.0), 20.0), ('noise d70', O.synth_noise(4, 120.0), 70.0),
                      ('tonal d20', O.synth_tonal(5, 120.0), 20.0), ('tonal d70', O.synth_tonal(6, 120.0), 70.0)):
    prm = O.Params(density=dens)
    stg = O.find_peaks_stages(d, prm)
    sg = stg['sgram']
    G = O.gauss_table(256, prm.f_sd)
    a = O.a_dec_of(dens)
    starts = list(range(200, sg.shape[1] - 1500, 97))
    r = conv_len(sg, a, G, prm.maxpksperframe, starts, 1500)
    ok = [x for x in r if x is not None]
    print(name, 'T', sg.shape[1], 'starts', len(r), 'converged', len(ok), 'warm-up frames: median', int(np.median(ok)), 'p90',
          int(np.percentile(ok, 90)), 'max', max(ok), 'not converged within 1500:', len(r) - len(ok))


def bwd_states(sgram, fwd, a_dec, G, t_end):
    """backward scan over frames t_end-1 .. 0 starting from the standard initialisation on column t_end-1; returns the
    state at ENTRY of each frame (dict frame -> thr)."""
    srows = sgram.shape[0]
    st = O.spreadpeaksinvector(sgram[:, t_end - 1], G)
    out = {}
    for col in range(t_end, 0, -1):
        out[col - 1] = st.copy()
        pk = np.nonzero(fwd[:, col - 1])[0]
        for val, p in sorted(zip(sgram[pk, col - 1], pk), reverse=True):
            if val >= st[p]:
                st = np.maximum(st, val * G[srows - p: 2 * srows - p])
        st = a_dec * st
    return out


print('backward:')
for name, d, dens in (('noise d20', O.synth_noise(3, 120.0), 20.0), ('noise d70', O.synth_noise(4, 120.0), 70.0),
                      ('tonal d20', O.synth_tonal(5, 120.0), 20.0), ('tonal d70', O.synth_tonal(6, 120.0), 70.0)):
    prm = O.Params(density=dens)
    stg = O.find_peaks_stages(d, prm)
    sg, fwd = stg['sgram'], stg['fwd']
    G = O.gauss_table(256, prm.f_sd)
    a = O.a_dec_of(dens)
    T = sg.shape[1]
    ref = bwd_states(sg, fwd, a, G, T)
    ws = []
    for te in range(1600, T - 100, 97):
        seg = bwd_states(sg[:, :te], fwd[:, :te], a, G, te)
        w = None
        for k in range(0, 1500):
            t = te - 1 - k
            if np.array_equal(seg[t], ref[t]):
                w = k
                break
        ws.append(w)
    ok = [x for x in ws if x is not None]
    print(name, 'ends', len(ws), 'converged', len(ok), 'median', int(np.median(ok)), 'p90', int(np.percentile(ok, 90)), 'max', max(ok))
