"""CPU, build container only: `--ncores N` of the UNCHANGED reference CLI reaches N GPUs through the drop-in.

The reference's workers -- `multiprocessing.Process` children of `audfprint.multiproc_add` (audfprint.py:199-235) and the
joblib / loky workers of `audfprint.do_cmd_multiproc` (audfprint.py:243-267) -- inherit one environment, so the drop-in
tells them apart by the ordinal their parent gave them (audfprint_amd/audfprint_analyze.py:_device).  Here the device count
is set to 8 (no GPU in this container) and the GPU call is replaced by a record of the device each worker chose
(tests/_ncores_helper.py); both of the reference's mechanisms run for real."""
import os
import sys
import textwrap

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')
REF_MODULES = ('audfprint', 'audfprint_match', 'hash_table', 'audio_read', 'stft', 'audfprint_analyze', 'docopt')


@pytest.fixture()
def ref_cli(monkeypatch, tmp_path):
    """The reference CLI module over the drop-in, installed the way INTEGRATION.md §1 says: a shim file named
    audfprint_analyze.py ahead of the reference on sys.path -- files, not sys.modules entries, so that joblib's freshly
    started workers resolve the same modules."""
    shim = tmp_path / 'shim'
    shim.mkdir()
    (shim / 'audfprint_analyze.py').write_text(textwrap.dedent('''\
        from audfprint_amd.audfprint_analyze import *          # noqa: F401,F403
        from audfprint_amd.audfprint_analyze import Analyzer, g2h_analyzer, extract_features_analyzer  # noqa: F401
        '''))
    (shim / 'docopt.py').write_text('def docopt(*a, **k):\n    return {}\n')      # not installed here; only main() calls it
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(here)
    monkeypatch.syspath_prepend(str(shim))
    monkeypatch.setenv('PYTHONPATH', os.pathsep.join([str(shim), here, root, REF]))
    for k in ('AFP_DEVICE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('AFP_DEVICE_COUNT', '8')
    for m in REF_MODULES:
        sys.modules.pop(m, None)
    import audfprint_amd.audfprint_analyze as M
    M._DEVICE_OF_PID.clear()
    import audfprint
    import audfprint_analyze
    assert audfprint_analyze.Analyzer is M.Analyzer and audfprint.audfprint_analyze is audfprint_analyze
    yield audfprint, M
    M._DEVICE_OF_PID.clear()
    for m in REF_MODULES:
        sys.modules.pop(m, None)


def _analyzer(tmp_path, nap=0.0):
    from _ncores_helper import RecordingAnalyzer
    a = RecordingAnalyzer()
    a.logdir = str(tmp_path / ('log%d' % len(list(tmp_path.iterdir()))))
    os.mkdir(a.logdir)
    a.nap = nap
    return a


@pytest.mark.parametrize('ncores, per_device', [(8, 1), (16, 2), (3, 1)])
def test_multiproc_add_spreads_its_children_over_the_gpus(ref_cli, tmp_path, ncores, per_device):
    """audfprint.multiproc_add (fork): child k of N opens GPU (k - 1) mod 8; the parent's own choice stays GPU 0."""
    audfprint, M = ref_cli
    from _ncores_helper import read_log
    import hash_table
    a = _analyzer(tmp_path)
    assert M._device() == 0                                  # the parent decides first (and caches it) ...
    ht = hash_table.HashTable(hashbits=10, depth=20, maxtime=16384)
    files = ['f%03d.wav' % i for i in range(3 * ncores)]
    said = []
    audfprint.multiproc_add(a, ht, iter(files), said.append, ncores)
    log = read_log(a.logdir)
    assert len(log) == ncores and sum(v[2] for v in log.values()) == len(files)
    devs = sorted(v[1] for v in log.values())
    want = sorted(k % 8 for k in range(ncores))
    assert devs == want, (devs, want)
    assert all(v[1] == (v[0] - 1) % 8 for v in log.values())           # device = (ordinal - 1) mod count
    assert max(np.bincount(devs)) == per_device
    assert M._device() == 0                                  # ... and its children did not move it
    # the merged table holds every file, and each file's rows name the device of the worker that had it (file i -> worker i % N)
    assert sorted(ht.names) == sorted(files) and int(ht.counts.sum()) == 3 * len(files)
    for i, fn in enumerate(files):
        rows = ht.retrieve(fn)
        assert sorted(int(h) % 100 for _, h in rows) == [((i % ncores)) % 8] * 3


def test_a_second_add_keeps_spreading(ref_cli, tmp_path):
    """The parent's process counter keeps running: the children of a second multiproc_add are (9,) .. (16,) -- still one per GPU."""
    audfprint, M = ref_cli
    from _ncores_helper import read_log
    import hash_table
    for _ in range(2):
        a = _analyzer(tmp_path)
        ht = hash_table.HashTable(hashbits=10, depth=20, maxtime=16384)
        audfprint.multiproc_add(a, ht, iter(['g%d.wav' % i for i in range(8)]), lambda m: None, 8)
        assert sorted(v[1] for v in read_log(a.logdir).values()) == list(range(8))


def test_joblib_precompute_spreads_its_workers_over_the_gpus(ref_cli, tmp_path):
    """audfprint.do_cmd_multiproc('precompute') (joblib, loky workers started from a clean interpreter)."""
    audfprint, M = ref_cli
    from _ncores_helper import read_log
    from joblib.externals.loky import get_reusable_executor
    get_reusable_executor(kill_workers=True).shutdown(wait=True)       # workers of an earlier test carry another environment
    a = _analyzer(tmp_path, nap=0.25)
    out = tmp_path / 'pre'
    out.mkdir()
    files = [str(tmp_path / ('p%02d.wav' % i)) for i in range(24)]
    said = []
    try:
        audfprint.do_cmd_multiproc('precompute', a, None, iter(files), None, str(out), 'hashes', said.append, ncores=8)
    finally:
        get_reusable_executor(kill_workers=True).shutdown(wait=True)
    log = read_log(a.logdir)
    assert sum(v[2] for v in log.values()) == len(files) and len(said) == len(files)
    assert all(v[0] >= 1 and v[1] == (v[0] - 1) % 8 for v in log.values())
    # loky starts its 8 workers with consecutive ordinals: the ones that took files sit on distinct GPUs
    devs = [v[1] for v in log.values()]
    assert len(log) >= 2 and len(set(devs)) == len(devs), log      # (how many of the 8 workers get a file depends on the host's load)
    assert M._device() == 0
    # every .afpt names the device of the worker that wrote it, through the reference's own writer path
    import audfprint_analyze
    seen = set()
    for fn in files:
        rows = audfprint_analyze.hashes_load(str(out) + os.path.splitext(fn)[0] + '.afpt')
        assert len(rows) == 3
        seen.add(rows[0][1] - 100)
    assert seen == set(devs)


def test_explicit_choices_win(ref_cli, monkeypatch):
    audfprint, M = ref_cli
    import multiprocessing

    def ask(q):
        q.put(M._device())

    def child_says():
        q = multiprocessing.Queue()
        p = multiprocessing.Process(target=ask, args=(q,))
        p.start()
        v = q.get(timeout=30)
        p.join()
        return v, p._identity[-1]
    v, k = child_says()
    assert v == (k - 1) % 8
    for env, val, want in (('AFP_DEVICE', '5', 5), ('AFP_DEVICE', 'first', 0), ('LOCAL_RANK', '3', 3)):
        monkeypatch.setenv(env, val)
        assert child_says()[0] == want
        M._DEVICE_OF_PID.clear()
        assert M._device() == want
        monkeypatch.delenv(env)
        M._DEVICE_OF_PID.clear()
    monkeypatch.setenv('AFP_DEVICE', '2')
    monkeypatch.setenv('LOCAL_RANK', '6')
    assert child_says()[0] == 2                               # AFP_DEVICE before LOCAL_RANK
    monkeypatch.setenv('AFP_DEVICE', 'auto')
    assert child_says()[0] == 6                               # auto: the launcher's LOCAL_RANK, else the ordinal
    monkeypatch.delenv('LOCAL_RANK')
    monkeypatch.setenv('AFP_DEVICE_COUNT', '2')
    v, k = child_says()
    assert v == (k - 1) % 2


def test_the_choice_is_made_once_per_process(ref_cli, monkeypatch):
    audfprint, M = ref_cli
    assert M._device() == 0
    monkeypatch.setenv('AFP_DEVICE', '4')
    assert M._device() == 0                                   # cached: a process does not hop between GPUs
    M._DEVICE_OF_PID.clear()
    assert M._device() == 4
