"""CPU: the bookkeeping of bench.py that does not need a GPU -- what a `parity` object says about WHICH rows were checked
(VERDICT r5 #1), the CPU-baseline switch between the oracle ("port") and the reference itself ("reference", AFP_REF_DIR),
the roofline object's arithmetic and its refusal of counters that belong to another build."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


@pytest.fixture(scope='module')
def bench():
    argv = sys.argv
    sys.argv = ['bench.py']
    try:
        spec = importlib.util.spec_from_file_location('afp_bench_under_test', os.path.join(ROOT, 'bench.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def _result(rows_per_clip, flags=None, path=None):
    from audfprint_amd.batch import BatchResult
    r = BatchResult()
    r.hash_offsets = np.concatenate([[0], np.cumsum([len(x) for x in rows_per_clip])]).astype(np.int64)
    r.hashes = np.concatenate(rows_per_clip).astype(np.int32).reshape(-1, 2) if rows_per_clip else np.zeros((0, 2), np.int32)
    r.unit_flags = np.zeros(len(rows_per_clip), np.int32) if flags is None else np.asarray(flags, np.int32)
    r.nclips = len(rows_per_clip)
    r.path = path or dict(compact=True, segments=False, redone_dense=False, near_tie_redone=False)
    return r


def test_parity_object_names_the_timed_rows(bench):
    from audfprint_amd import _lib
    B = types.SimpleNamespace(lib=_lib)
    a = [np.array([[1, 2], [3, 4]]), np.array([[5, 6]])]
    timed, guard = _result(a), _result(a, flags=[0, _lib.UNIT_NEARTIE])
    p = bench.timed_parity(B, timed, guard, True, 2, 'row by row')
    assert p['timed_variant_checked'] is True and p['bit_exact'] is True and p['clips_checked'] == 2
    assert 'LAST TIMED step' in p['how'] and 'guard off' in p['how']
    assert p['guarded_pass_identical'] is True and p['near_tie_units'] == 1 and p['near_tie_eps'] == bench.NT_EPS
    assert p['timed_path'] == dict(compact=True, segments=False, redone_dense=False) and p['tie_prone_units'] == 0
    other = _result([a[0], np.array([[5, 7]])])
    assert bench.timed_parity(B, timed, other, True, 2, 'x')['guarded_pass_identical'] is False
    assert 'near_tie_units' not in bench.timed_parity(B, timed, None, False, 2, 'x')
    assert bench.gpu_digests(timed, [0, 1]) == [(2, bench._digest(a[0])), (1, bench._digest(a[1]))]
    assert bench.same_rows(timed, _result(a)) and not bench.same_rows(timed, other)


def test_cpu_rows_are_the_oracles_and_under_afp_ref_dir_the_references(bench, monkeypatch):
    from oracle import afp_oracle as O
    kw = dict(density=70.0, maxpairsperpeak=10, shifts=4)
    d = O.synth_noise(3, 4.0)
    monkeypatch.delenv('AFP_REF_DIR', raising=False)
    f, kind = bench.cpu_rows_fn(O, kw)
    assert kind == 'port' and bench.reference_tree() is None
    want = O.extract(d, O.Params(**kw))[1]
    assert np.array_equal(f(d), want)
    monkeypatch.setenv('AFP_REF_DIR', '/nonexistent')
    assert bench.cpu_rows_fn(O, kw)[1] == 'port'                       # no tree there: never looked for anywhere else
    if os.path.isdir(REF):
        monkeypatch.setenv('AFP_REF_DIR', REF)
        g, kind = bench.cpu_rows_fn(O, kw)
        assert kind == 'reference' and np.array_equal(g(d), want)
        assert np.array_equal(bench.cpu_rows_fn(O, dict(density=20.0, maxpairsperpeak=3, shifts=1))[0](d), O.extract(d, O.Params())[1])
        assert len(bench.reference_rows(REF, np.zeros(4000, np.float32), dict())) == 0


def test_roofline_object_and_stale_counters(bench, monkeypatch, tmp_path):
    wl = bench.WORKLOADS['c3']
    nclips, nsamp, nh = 1024, 330750, 2014801
    kern = {'k_stft': 1.0, 'k_scan': 0.8, 'pipeline(first launch..last launch)': 2.0}
    bid = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))['c3_build_id']
    r = bench.roofline_obj('c3', wl, nclips, nsamp, nh, 1.5, kern, 2000.0, bid)
    alg = 4.0 * nclips * nsamp + 8.0 * nh
    assert r['alg_bytes_per_launch'] == alg == 1370870408.0 and r['kernel'] == 'k_stft' and r['peak'] == 8000.0
    assert abs(r['achieved'] - alg / 1e-3 / 1e9) < 0.01 and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-5
    assert abs(r['whole_step_frac'] - alg / 1.5e-3 / 1e9 / 8000.0) < 1e-5
    assert r['traffic'] and r['traffic_over_algorithmic'] > 1.0 and r['profile_build_id'] == bid and r['bound'] == 'hbm' and r['limited_by'] in ('hbm', 'valu_issue')
    stale = bench.roofline_obj('c3', wl, nclips, nsamp, nh, 1.5, kern, 2000.0, 'someotherbuild00')
    assert stale['traffic'] is None and 'stale_profile' in stale and 'valu_issue' not in stale and stale['bound'] == 'hbm'
    assert bench.frames_of(330750, 1) == 1292 and bench.frames_of(330750, 4) == sum(1 + (330750 - o) // 256 for o in (0, 64, 128, 192))


def test_cgroup_limit_is_a_number_or_none(bench, monkeypatch):
    v = bench.cgroup_cpu_limit()
    assert v is None or v > 0
    n = bench.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    monkeypatch.setattr(bench, 'cgroup_cpu_limit', lambda: 2.5)
    assert bench.effective_cpus() == min(3, os.cpu_count() or 1, len(os.sched_getaffinity(0)))
