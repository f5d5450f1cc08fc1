#!/usr/bin/env python
"""How long does the onset filter need to FORGET a few-ulp difference in its state, bit for bit?  (k_hpf chunk mode, round 6.)

The recurrence  y = x + z ; z = -x + 0.98 y  (audfprint_analyze.py:293-295, lfilter's direct form II transposed) contracts by
0.98 per frame, so two runs over the same rows that start a few ulps apart end up on the SAME float64 bit pattern -- after how
many frames?  For each warm-up length W: approximate states every W frames (zero-state end states of the pieces, folded
linearly: what pass 1 + the fold of k_hpf produce, a few 1e-15 off), the exact recurrence over the next W frames, and the count
of bins that still differ from the sequential filter's state at the end.  numpy, oracle spectrograms; takes a few minutes.

    python tools/hpf_merge.py > profiles/r06_hpf_merge.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import afp_oracle as O          # noqa: E402

POLE = 0.98


def run(x, z, a, b, rec=None):
    for n in range(a, b):
        xn = x[:, n]
        yn = xn + z
        z = xn * (-1.0) - yn * (-POLE)
        if rec is not None and (n + 1) % 64 == 0:
            rec[n + 1] = z.copy()
    return z


def main():
    tot = {256: [0, 0], 384: [0, 0], 512: [0, 0], 640: [0, 0], 768: [0, 0], 1024: [0, 0]}
    clips = ([('noise', O.synth_noise(100 + i, 300.0)) for i in range(6)] + [('tonal', O.synth_tonal(200 + i, 200.0)) for i in range(3)] +
             [('noise-silence-noise', np.concatenate([O.synth_noise(300, 100.0), np.zeros(11025 * 20, np.float32), O.synth_noise(301, 100.0)]))])
    worst = 0.0
    for name, d in clips:
        logs, _ = O.log_sgram(O.stft_complex(d))
        x = np.ascontiguousarray(logs[:-1, :])
        T = x.shape[1]
        truth = {0: np.zeros(256)}
        run(x, np.zeros(256), 0, T, truth)
        for W in tot:
            nb = T // W
            zt = [np.zeros(256)]
            pc = POLE ** W
            for k in range(nb):
                zt.append(run(x, np.zeros(256), k * W, (k + 1) * W) + pc * zt[-1])
            worst = max(worst, max(float(np.abs(zt[k] - truth[k * W]).max()) for k in range(nb + 1)))
            for k in range(1, nb):
                s = run(x, zt[k - 1], (k - 1) * W, k * W)
                tot[W][0] += int(np.count_nonzero(s != truth[k * W]))
                tot[W][1] += 256
        print('%-20s %6d frames | bins still apart / bins compared after W frames: %s' % (name, T, {W: tuple(v) for W, v in tot.items()}), flush=True)
    print('largest error of a folded (approximate) state: %.2e' % worst)
    print('cumulative fraction apart: %s' % {W: '%.1e' % (v[0] / max(1, v[1])) for W, v in tot.items()})


if __name__ == '__main__':
    main()
