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
           '--nclips', '96', '--secs', '10', '--pool', '96', '--c4-clips', '600', '--c4-batch', '200']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['clips_per_gpu'] == 96
    assert d['parity']['bit_exact'] is True and d['parity']['ranks'] == 2 and d['parity']['clips_checked_per_rank'] == 64
    # the rows checked are those of the LAST TIMED step (guard off); the guarded pass only counts near ties and must agree
    assert d['parity']['timed_variant_checked'] is True and d['parity']['guarded_pass_identical'] is True
    # every rank reports where it ran and what it was bound to
    assert sorted(r['rank'] for r in d['ranks_seen']) == [0, 1]
    for r in d['ranks_seen']:
        assert 'numa_node' in r and 'cpus_bound' in r and r['cpus_allowed'] >= 1
    # the host-inclusive leg runs on every rank at the same time: per-rank and aggregate link rates
    hp = d['host_inclusive_pipelined']
    assert hp['ranks'] == 2 and len(hp['per_rank']) == 2 and all(r['pcie_gb_per_s'] > 0 for r in hp['per_rank'])
    assert abs(hp['aggregate_pcie_gb_per_s'] - sum(r['pcie_gb_per_s'] for r in hp['per_rank'])) < 0.2
    # BASELINE configs[3] as a job, per rank and aggregate, each rank's table bit-exact against the oracle's HashTable
    cj = d['c4_job']
    assert cj['ranks'] == 2 and len(cj['per_rank']) == 2 and cj['bit_exact'] is True, cj
    for j in cj['per_rank']:
        assert j['clips'] == 600 and j['batches'] == 3 and j['parity']['bit_exact'] is True and j['parity']['clips_checked'] == 200
        assert all(j['invariants'].values()), j['invariants']
        assert set(j['stages_ms']) >= {'table_store_kernels', 'overflow_replay', 'download_to_host_arrays', 'waiting_for_batches'}
    assert cj['aggregate_hashes_per_s'] > 0 and cj['aggregate_pcie_gb_per_s'] > 0
    m = d['table_merge_across_ranks']
    assert 'error' not in m, m
    assert m['ranks'] == 2 and m['merged_ids'] == 1200 and m['counts_add_up'] is True and m['counts_equal_reference_rule'] is True
    assert m['hashes_stored_all_ranks'] == sum(j['hashes'] for j in cj['per_rank'])
    assert m['table_total_count'] == m['hashes_stored_all_ranks'] - m['counts_clipped_to_depth_on_rank0'] - m['counts_clipped_on_the_way_in_other_ranks']
    # what crosses to rank 0 is the PACKED table -- counts + filled row prefixes -- not 424 MB per rank
    assert m['transport'] == 'staged' and m['fallback'] is None                     # (gloo: through the host)
    assert 0 < m['bytes_per_sending_rank'] * 10 <= m['dense_table_bytes_per_rank'], m
    assert m['bytes_per_sending_rank'] == 4 * ((1 << 20) + cj['per_rank'][1]['hashes'] - 0) or m['bytes_per_sending_rank'] <= 4 * ((1 << 20) + cj['per_rank'][1]['hashes'])


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


@pytest.mark.gpu
def test_the_line_survives_a_crash_in_the_multi_rank_extras():
    """The N > 1 extras exercise transports no one-GPU box can run; if rank 0 dies in them -- here: SIGSEGV right before the
    host-inclusive leg -- the guard process prints the contract line as it stood (headline + per-rank parity), marked."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', AFP_BENCH_ONE_GPU='1', AFP_BENCH_BACKEND='gloo', AFP_BENCH_CRASH_IN_EXTRAS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29643', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1',
           '--nclips', '64', '--secs', '5', '--pool', '64', '--c4-clips', '200', '--c4-batch', '100']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode != 0                                   # rank 0 did die
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(lines[0])
    assert 'extras_crashed' in d and d['n_gpus'] == 2 and d['value'] > 0 and d['steps'] == 4
    assert d['parity']['bit_exact'] is True and d['parity']['timed_variant_checked'] is True
    assert 'c4_job' not in d and 'host_inclusive_pipelined' not in d
