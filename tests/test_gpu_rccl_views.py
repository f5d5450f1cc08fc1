"""what ONE GPU can prove about the RCCL leg of the sharded job (audfprint.py:217-235 as
shard.merge_tables_to_rank0): a world-size-1 `nccl` process group runs collectives ON views of the library's own
hipMalloc memory (table counts, packed values -- torch.as_tensor over __cuda_array_interface__, no copy) and on a
torch-owned receive buffer that afp_table_merge_packed_device then reads; the merged table equals the live reference's
golden.  Point-to-point between two GPUs stays unproven on this box (one GPU per call)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, random
    sys.path.insert(0, %(root)r)
    import numpy as np
    import audfprint_amd
    audfprint_amd.configure_runtime()
    import torch
    import torch.distributed as dist
    from audfprint_amd import shard
    from audfprint_amd.batch import Extractor
    from audfprint_amd.table import TableBuilder
    from oracle import afp_oracle as O
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend='nccl', init_method='tcp://127.0.0.1:%(port)d', rank=0, world_size=1)
    z = np.load(os.path.join(%(root)r, 'tests', 'golden', 'table_merge.npz'))
    names = [str(n) for n in z['names']]
    off, nsplit = z['offsets'], int(z['nsplit'])
    tag, hbits, da, db = 's', 10, 4, 4
    nb = 1 << hbits
    a, b = O.OracleHashTable(hashbits=hbits, depth=da), O.OracleHashTable(hashbits=hbits, depth=db)
    ta, tb = TableBuilder(a, Extractor.get(0)), TableBuilder(b, Extractor(0))
    random.seed(11)
    ta.store_batch(names[:nsplit], rows=z['rows'][:off[nsplit]], offsets=off[:nsplit + 1])
    random.seed(12)
    tb.store_batch(names[nsplit:], rows=z['rows'][off[nsplit]:], offsets=off[nsplit:] - off[nsplit])
    # 1. the probe the device transport starts with
    assert shard._alias_probe(tb, dev)
    # 2. collectives on views of LIBRARY memory (the sender's side of shard.py): counts and packed values
    n = tb.pack()
    vp, cp, n2 = tb.packed_device_ptrs()
    assert n == n2 > 0
    torch.cuda.synchronize(dev)
    vv = torch.as_tensor(shard._DevMem(vp, n * 4), device=dev).view(torch.int32)
    vc = torch.as_tensor(shard._DevMem(cp, nb * 4), device=dev).view(torch.int32)
    assert vv.data_ptr() == vp and vc.data_ptr() == cp
    keep_v, keep_c = vv.clone(), vc.clone()
    dist.all_reduce(vc)                      # (sum over one rank: the identity -- RCCL has read and written the library's buffer)
    dist.broadcast(vv, src=0)
    w = dist.all_reduce(vv, op=dist.ReduceOp.MAX, async_op=True)
    w.wait()
    torch.cuda.synchronize(dev)
    assert torch.equal(vv, keep_v) and torch.equal(vc, keep_c)
    # 3. the receiver's side: torch-owned buffers filled by a collective, handed to the library as raw pointers
    bv = torch.zeros(n, dtype=torch.int32, device=dev)
    bc = torch.zeros(nb, dtype=torch.int32, device=dev)
    bv.copy_(vv)
    bc.copy_(vc)
    dist.broadcast(bv, src=0)
    dist.broadcast(bc, src=0)
    torch.cuda.synchronize(dev)              # (shard.py: the merge is queued on the library's stream only after this)
    np.random.seed(4321)
    nov = ta.merge(shard._RemoteTable(b.names, b.hashesperid, db, b.maxtimebits), other_device_ptrs=(bv.data_ptr(), bc.data_ptr()), packed=True)
    ta.finalize()
    assert np.array_equal(a.counts, z[tag + '_m_counts']) and np.array_equal(a.table, z[tag + '_m_table'])
    assert np.array_equal(a.hashesperid, z[tag + '_m_hpi']) and a.names == [str(x) for x in z[tag + '_m_names']]
    # 4. the whole function at world size 1 is the reference's --ncores 1: nothing merged, nothing clipped
    st = {}
    assert shard.merge_tables_to_rank0(tb, dist, dev, stats=st) == [] and st['transport'] is None
    dist.destroy_process_group()
    print('rccl views ok, merged with', nov, 'over-full buckets')
''')


@pytest.mark.gpu
def test_rccl_collectives_on_library_memory_world_size_1(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(WORKER % dict(root=ROOT, port=29671))
    env = dict(os.environ, AFP_BACKTRACE='1', PYTHONFAULTHANDLER='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'rccl views ok' in out.stdout
