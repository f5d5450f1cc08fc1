"""CPU, world_size 2 over gloo: clip sharding covers every clip exactly once and the job
statistics reduce as bench.py expects (max elapsed, summed hashes / audio seconds)."""
import os
import subprocess
import sys
import textwrap

from audfprint_amd.shard import shard_bounds, shard_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partitions_are_exact():
    for n in (0, 1, 7, 8, 1024, 100000):
        for world in (1, 2, 3, 8):
            rr = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for s in rr for i in s) == list(range(n))
            bb = [shard_bounds(n, r, world) for r in range(world)]
            assert bb[0][0] == 0 and bb[-1][1] == n
            assert all(bb[r][1] == bb[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in bb]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from audfprint_amd.shard import shard_indices, reduce_job_stats, all_ranks_true
    from oracle import afp_oracle as O
    dist.init_process_group(backend='gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    nclips = 5
    mine = shard_indices(nclips, rank, world)
    # each rank fingerprints its own clips (CPU oracle stands in for the GPU path here)
    nh = sum(len(O.extract(O.synth_noise(100 + i, 1.0))[1]) for i in mine)
    el, th, ta = reduce_job_stats(1.0 + rank, nh, 1.0 * len(mine), dist, None)
    want = sum(len(O.extract(O.synth_noise(100 + i, 1.0))[1]) for i in range(nclips))
    assert el == float(world) and th == float(want) and ta == float(nclips), (el, th, ta, want)
    # the per-rank parity verdicts of bench.py are AND-ed over the ranks
    assert all_ranks_true(True, dist, None) is True
    assert all_ranks_true(rank != 1, dist, None) is False
    dist.barrier()
    dist.destroy_process_group()
    print('rank', rank, 'ok')
''')


def test_two_rank_gloo_job(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29617', str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('ok') == 2
