"""Run in a FRESH interpreter on a GPU box (tests/test_gpu_dropin.py::test_ncores_2_on_one_gpu): `--ncores 2` of the
reference CLI over the drop-in on ONE GPU -- both of the reference's mechanisms, `multiprocessing.Process` children (new,
audfprint.py:199-235) and joblib workers (precompute, audfprint.py:243-267), against the same files handled by one process.

With a reference tree (AFP_REF_DIR) the REAL audfprint.multiproc_add / do_cmd_multiproc / do_cmd drive it, unchanged; on the
GPU box, which has none, stand-ins issue the same calls in the same order.  The parent touches the GPU only after its
children have finished (a HIP context does not survive fork)."""
import multiprocessing
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NCORES = 2
LINES = []


def say(*a):
    s = ' '.join(str(x) for x in a)
    LINES.append(s)
    print(s, flush=True)


def _standin_worker(analyzer, files, hashbits, depth, maxtime, tx):
    """make_ht_from_list of a worker (audfprint.py:137-146): a private table, ingest() per file, the table sent back"""
    from _ncores_helper import StandInTable
    ht = StandInTable(hashbits, depth, maxtime)
    for fn in files:
        analyzer.ingest(ht, fn)
    tx.send(ht)
    tx.close()


def standin_multiproc_add(analyzer, table, files, report, ncores):
    lists = [files[k::ncores] for k in range(ncores)]
    rx, pr = [], []
    for k in range(ncores):
        r, t = multiprocessing.Pipe(False)
        p = multiprocessing.Process(target=_standin_worker, args=(analyzer, lists[k], table.hashbits, table.depth,
                                                                 1 << table.maxtimebits, t))
        p.start()
        rx.append(r)
        pr.append(p)
    for k in range(ncores):
        part = rx[k].recv()
        report(['hash_table %d has %d files %d hashes' % (k, len(part.names), int(part.counts.sum()))])
        table.merge(part)
        pr[k].join()


def _standin_precompute(analyzer, fn, outdir):
    """file_precompute (audfprint.py:70-116): wavfile2hashes + hashes_save under outdir"""
    import audfprint_analyze
    rel = '/'.join(c for c in fn.split('/') if c not in ('.', '..', ''))
    out = os.path.join(outdir, os.path.splitext(rel)[0] + audfprint_analyze.PRECOMPEXT)
    h = analyzer.wavfile2hashes(fn)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    audfprint_analyze.hashes_save(out, h)
    return ['wrote %s ( %d hashes, %.3f sec)' % (out, len(h), analyzer.soundfiledur)]


def main():
    tmp = tempfile.mkdtemp(prefix='afp_ncores_')
    shim = os.path.join(tmp, 'shim')
    os.mkdir(shim)
    with open(os.path.join(shim, 'audfprint_analyze.py'), 'w') as f:
        f.write('from audfprint_amd.audfprint_analyze import *          # noqa: F401,F403\n'
                'from audfprint_amd.audfprint_analyze import Analyzer, g2h_analyzer, extract_features_analyzer  # noqa: F401\n')
    with open(os.path.join(shim, 'docopt.py'), 'w') as f:                       # not installed; only audfprint.main() calls it
        f.write('def docopt(*a, **k):\n    return {}\n')
    with open(os.path.join(shim, 'audio_read.py'), 'w') as f:                   # ffmpeg is not installed anywhere here
        f.write('import numpy as np, scipy.io.wavfile\n'
                'def audio_read(filename, sr=None, channels=None):\n'
                '    rate, w = scipy.io.wavfile.read(filename)\n'
                '    return w.astype(np.float32) / np.float32(32768), rate      # audio_read.buf_to_float, audio_read.py:121-145\n')
    ref = os.environ.get('AFP_REF_DIR', '').strip()
    have_ref = bool(ref) and os.path.isfile(os.path.join(ref, 'audfprint.py'))
    path = [shim, HERE, ROOT] + ([ref] if have_ref else [])
    sys.path[:0] = path
    os.environ['PYTHONPATH'] = os.pathsep.join(path)                            # joblib's workers start from a clean interpreter
    for k in ('AFP_DEVICE', 'LOCAL_RANK', 'AFP_DEVICE_COUNT'):
        os.environ.pop(k, None)

    import scipy.io.wavfile
    import audfprint_amd.audfprint_analyze as M
    from _ncores_helper import LoggingAnalyzer, StandInTable, read_log, rows_of
    from oracle import afp_oracle as O
    files = []
    for i in range(8):
        fn = os.path.join(tmp, 'clip%d.wav' % i)
        scipy.io.wavfile.write(fn, 11025, np.round(O.synth_noise(4400 + i, 4.0 + 0.5 * i) * 32768).astype(np.int16))
        files.append(fn)
    an = LoggingAnalyzer()
    an.density, an.maxpairsperpeak, an.shifts = 20.0, 3, 1
    say('reference CLI tree: %s' % (ref if have_ref else 'absent -- stand-ins issue the same calls (tests/_ncores_gpu_run.py)'))

    # ---- new --ncores 2: two forked children, each ingests its half into a private table, the parent merges ------------
    an.logdir = os.path.join(tmp, 'log_new')
    os.mkdir(an.logdir)
    reports = []
    np.random.seed(0)
    if have_ref:
        import audfprint
        import hash_table
        ht2 = hash_table.HashTable(hashbits=20, depth=100, maxtime=16384)
        audfprint.multiproc_add(an, ht2, iter(files), reports.extend, NCORES)
    else:
        ht2 = StandInTable(20, 100, 16384)
        standin_multiproc_add(an, ht2, files, reports.extend, NCORES)
    for r in reports:
        say('  new --ncores %d:' % NCORES, r)
    log_new = read_log(an.logdir)
    say('  new --ncores %d: workers (pid: ordinal, device) %s' % (NCORES, {p: v[:2] for p, v in sorted(log_new.items())}))

    # ---- precompute --ncores 2: joblib workers ---------------------------------------------------------------------------
    an.logdir = os.path.join(tmp, 'log_pre')
    os.mkdir(an.logdir)
    out2 = os.path.join(tmp, 'pre2')
    reports = []
    if have_ref:
        audfprint.do_cmd_multiproc('precompute', an, None, iter(files), None, out2, 'hashes', reports.extend, ncores=NCORES)
    else:
        import joblib
        for msgs in joblib.Parallel(n_jobs=NCORES)(joblib.delayed(_standin_precompute)(an, fn, out2) for fn in files):
            reports.extend(msgs)
    log_pre = read_log(an.logdir)
    say('  precompute --ncores %d: %d files written; workers (pid: ordinal, device) %s'
        % (NCORES, len(reports), {p: v[:2] for p, v in sorted(log_pre.items())}))

    # ---- the same files by ONE process (--ncores 1): the parent, which touches the GPU only now --------------------------
    an.logdir = os.path.join(tmp, 'log_one')
    os.mkdir(an.logdir)
    out1 = os.path.join(tmp, 'pre1')
    if have_ref:
        ht1 = hash_table.HashTable(hashbits=20, depth=100, maxtime=16384)
        audfprint.do_cmd('new', an, ht1, iter(files), None, None, None, lambda m: None)
        audfprint.do_cmd('precompute', an, None, iter(files), None, out1, 'hashes', lambda m: None)
    else:
        ht1 = StandInTable(20, 100, 16384)
        for fn in files:
            an.ingest(ht1, fn)
            _standin_precompute(an, fn, out1)
    log_one = read_log(an.logdir)

    # ---- verdicts ---------------------------------------------------------------------------------------------------------
    ok = True
    want = [O.extract(O.synth_noise(4400 + i, 4.0 + 0.5 * i), O.Params())[1] for i in range(8)]
    rel = lambda fn: '/'.join(c for c in fn.split('/') if c not in ('.', '..', ''))
    nrows = 0
    for fn, h in zip(files, want):
        a = open(os.path.join(out2, os.path.splitext(rel(fn))[0] + '.afpt'), 'rb').read()
        b = open(os.path.join(out1, os.path.splitext(rel(fn))[0] + '.afpt'), 'rb').read()
        ok = ok and a == b == b'audfprinthashV00' + h.astype('<i4').tobytes()
        r2, r1 = rows_of(ht2, fn), rows_of(ht1, fn)
        ok = ok and r2 == r1 == sorted(zip((h[:, 0] & 16383).tolist(), (h[:, 1] & 0xFFFFF).tolist()))
        nrows += len(r1)
    say('  rows of --ncores %d == rows of --ncores 1 == oracle, file by file (.afpt bytes and table entries): %s (%d rows, %d files)'
        % (NCORES, ok, nrows, len(files)))
    ok_names = sorted(ht2.names) == sorted(ht1.names) == sorted(files) and int(ht2.counts.sum()) == int(ht1.counts.sum()) == nrows
    say('  merged table: %d names, %d entries; single-process table: %d names, %d entries: %s'
        % (len(ht2.names), int(ht2.counts.sum()), len(ht1.names), int(ht1.counts.sum()), ok_names))
    ndev = M._device_count()
    devs_ok = (len(log_new) == NCORES and all(v[0] >= 1 and v[1] == (v[0] - 1) % ndev for v in log_new.values())
               and 1 <= len(log_pre) <= NCORES and all(v[0] >= 1 and v[1] == (v[0] - 1) % ndev for v in log_pre.values())
               and len(log_one) == 1 and list(log_one.values())[0][:2] == (0, 0) and list(log_one) == [os.getpid()])
    say('  %d GPU(s) visible: every worker opened GPU (ordinal - 1) mod %d, the parent (ordinal 0) GPU 0: %s' % (ndev, ndev, devs_ok))
    good = ok and ok_names and devs_ok
    say('NCORES2 %s' % ('OK' if good else 'FAILED'))
    outdir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, 'r06_ncores2_on_one_gpu.log'), 'w') as f:
            f.write('\n'.join(LINES) + '\n')
    return 0 if good else 1


if __name__ == '__main__':
    sys.exit(main())
