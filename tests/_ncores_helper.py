"""Importable by worker processes of the --ncores tests (fork children and joblib / loky workers alike): an
Analyzer of the drop-in module whose GPU call is replaced by a record of WHICH device the worker would open."""
import os
import time

import numpy as np

import audfprint_amd.audfprint_analyze as M


class RecordingAnalyzer(M.Analyzer):
    """wavfile2hashes writes `<pid> <worker ordinal> <device>` into `self.logdir` and returns three rows that name
    the device, instead of asking the GPU.  The instance pickles into joblib workers like the real Analyzer."""
    logdir = None
    nap = 0.0

    def wavfile2hashes(self, filename):
        dev = M._device()                         # what Analyzer._extractor() asks before every launch
        with open(os.path.join(self.logdir, '%d.%s.dev' % (os.getpid(), os.path.basename(filename))), 'w') as f:
            f.write('%d %d %d\n' % (os.getpid(), M._worker_ordinal(), dev))
        if self.nap:
            time.sleep(self.nap)                  # (long enough for the pool to hand every worker a file)
        self.soundfiledur = 1.0
        self.soundfiletotaldur += 1.0
        self.soundfilecount += 1
        return np.array([[1, 100 + dev], [2, 200 + dev], [3, 300 + dev]], dtype=np.int32)


def read_log(logdir):
    """{pid: (ordinal, device, files)} of every worker that wrote into logdir."""
    out = {}
    for fn in sorted(os.listdir(logdir)):
        if fn.endswith('.dev'):
            with open(os.path.join(logdir, fn)) as f:
                pid, k, dev = (int(x) for x in f.read().split())
            prev = out.get(pid)
            assert prev is None or prev[:2] == (k, dev), 'a worker changed its device between files'
            out[pid] = (k, dev, (prev[2] if prev else 0) + 1)
    return out
