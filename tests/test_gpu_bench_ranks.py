"""GPU: the N > 1 logic of bench.py (per-rank parity AND-ed over the ranks, whole-job statistics, the cross-rank
HashTable.merge of the sharded `new -> fpdbase` job, exactly one JSON line on stdout) run as two torchrun ranks that
share the one GPU of the test box, collectives over gloo (RCCL needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', AFP_BENCH_ONE_GPU='1', AFP_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29641', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
           '--nclips', '96', '--secs', '10', '--pool', '96']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['clips_per_gpu'] == 96
    assert d['parity']['bit_exact'] is True and d['parity']['ranks'] == 2 and d['parity']['clips_checked_per_rank'] == 8
    m = d['table_merge_across_ranks']
    assert 'error' not in m, m
    assert m['ranks'] == 2 and m['merged_ids'] == 192 and m['counts_add_up'] is True
    assert m['table_total_count'] == m['hashes_stored_all_ranks'] == d['hashes_per_step']


@pytest.mark.gpu
def test_plain_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun starts the two ranks itself (it used to measure one GPU and print
    n_gpus: 1): one JSON line, n_gpus == 2, both ranks listed in ranks_seen."""
    env = dict(os.environ, AFP_BENCH_ONE_GPU='1', AFP_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1',
           '--nclips', '64', '--secs', '5', '--pool', '64', '--no-table']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2
    assert sorted(r['rank'] for r in d['ranks_seen']) == [0, 1]
    assert d['parity']['bit_exact'] is True and d['parity']['ranks'] == 2


@pytest.mark.gpu
def test_gpus_beyond_the_box_fails_loudly():
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'AFP_BENCH_ONE_GPU'):
        env.pop(k, None)
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0
    assert 'GPU(s) visible' in (out.stderr + out.stdout)
    assert not [ln for ln in out.stdout.splitlines() if ln.strip().startswith('{')]
