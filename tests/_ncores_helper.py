"""Importable by worker processes of the --ncores tests (fork children and joblib / loky workers alike): an
Analyzer of the drop-in module whose GPU call is replaced by a record of WHICH device the worker would open."""
import os
import time

import numpy as np

import audfprint_amd.audfprint_analyze as M
from oracle import afp_oracle as _O          # (test infrastructure: the stand-in table of the GPU-box harness)


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


class LoggingAnalyzer(M.Analyzer):
    """The REAL drop-in Analyzer (GPU calls and all) that also writes down which device each process opened."""
    logdir = None

    def _extractor(self, shifts):
        with open(os.path.join(self.logdir, '%d.any.dev' % os.getpid()), 'w') as f:
            f.write('%d %d %d\n' % (os.getpid(), M._worker_ordinal(), M._device()))
        return M.Analyzer._extractor(self, shifts)


def rows_of(table, name):
    """The (time, hash) rows a HashTable-like object (reference HashTable or OracleHashTable: table, counts, names,
    maxtimebits) holds for `name`, sorted -- hash_table.py:117-123 read backwards."""
    idv = table.names.index(name) + 1
    tb = int(table.maxtimebits)
    depth = table.table.shape[1]
    filled = np.arange(depth)[None, :] < np.minimum(table.counts, depth)[:, None]
    b, s = np.nonzero(filled & ((table.table >> tb) == idv))
    t = (table.table[b, s] & ((1 << tb) - 1)).astype(np.int64)
    return sorted(zip(t.tolist(), b.tolist()))


class StandInTable(_O.OracleHashTable):
    """What stands in for hash_table.HashTable where the reference tree is absent (the GPU box): the oracle's restatement
    behind the reference's two-argument store() (hash_table.py:91) and one-argument merge() (:291)."""

    def store(self, name, hashes):
        import random
        return _O.OracleHashTable.store(self, name, np.asarray(hashes, dtype=np.int32).reshape(-1, 2), random)

    def merge(self, other):
        return _O.OracleHashTable.merge(self, other, np.random)
