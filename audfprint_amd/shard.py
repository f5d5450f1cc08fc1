"""File-sharded multi-GPU ingest: one process per GPU, clips partitioned across ranks, NO
data-path collective (every clip's peaks/hashes depend on that clip only -- SURVEY.md §8e).
torch.distributed (RCCL on GPUs, gloo on CPU) is used only for the barrier and for reducing
the job statistics (elapsed = max over ranks, hashes/audio = sum over ranks)."""


def shard_indices(n_items, rank, world):
    """Round-robin partition, the reference's own rule for --ncores workers
    (audfprint.py:211-214: filelists[ix % ncores])."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError('bad rank/world')
    return list(range(rank, n_items, world))


def shard_bounds(n_items, rank, world):
    """Contiguous balanced partition [lo, hi) (used when the batch is one packed PCM buffer)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError('bad rank/world')
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_job_stats(elapsed, hashes, audio_sec, dist=None, device=None):
    """(max elapsed, sum hashes, sum audio seconds) over all ranks; identity when dist is None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed), float(hashes), float(audio_sec)
    import torch
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(hashes), float(audio_sec)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t[0].item()), float(s[0].item()), float(s[1].item())


def all_ranks_true(flag, dist=None, device=None):
    """AND of a per-rank boolean over all ranks (each rank checks its own shard against the oracle)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    import torch
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t[0].item() > 0.5)


class _RemoteTable(object):
    """What HashTable.merge reads of the other table besides its arrays (hash_table.py:295-299)."""

    def __init__(self, names, hashesperid, depth, maxtimebits, table=None, counts=None):
        self.names, self.hashesperid = list(names), hashesperid
        self.depth, self.maxtimebits = int(depth), int(maxtimebits)
        self.table, self.counts = table, counts


class _DevMem(object):
    """A raw device allocation presented to torch without a copy (`torch.as_tensor` reads __cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = dict(shape=(int(nbytes),), typestr='|u1', data=(int(ptr), False), version=2)


def merge_tables_to_rank0(tb, dist, device=None, fresh_parent=True):
    """The ONE exchange step of the sharded `new -> fpdbase` job (BASELINE configs[3]): every rank has built a private
    table from its clips (audfprint.py:204-224); the parent merges the workers' tables in worker order with
    HashTable.merge (audfprint.py:226-235, hash_table.py:291-323).  Here rank 0 is the parent: ranks 1..N-1 ship their
    tables and rank 0 merges them in rank order (same ids / names order as the reference's loop; over-full buckets draw
    np.random.permutation on rank 0 exactly as there).

    Transport: with RCCL ("nccl") the table and counts arrays go GPU to GPU (point-to-point over xGMI) straight out of
    the sender's table memory into a receive buffer that `afp_table_merge_device` reads -- no host round trip; the receive
    of rank r+1 is posted before rank r's merge starts.  With gloo (CPU tests, or two processes sharing one GPU) the
    arrays are staged through the host.  `tb` is this rank's audfprint_amd.table.TableBuilder; returns, on rank 0, the
    list of over-full bucket counts per merged rank (None elsewhere).

    `fresh_parent` (default), world size > 1 only: the reference's parent starts EMPTY for `new` (audfprint.py:436-443, 226-235)
    and merges every worker's table into it -- core 0's included -- which clips `counts[k]` of core 0's over-full buckets to
    `depth` on the way in (hash_table.py:304-305, 315-321 with an empty self: len(allvals) = min(count, depth)).  Rank 0's own
    table is the merge base here, so rank 0 first clips its counts the same way (`TableBuilder.clip_counts`, one tiny kernel)
    and the result -- table rows, counts, names, hashesperid, np.random draws -- is the reference parent's, bit for bit
    (golden from the reference's own loop: tests/golden/table_multiproc.npz).  With fresh_parent=False rank 0's table is
    taken as an already populated parent (the `add` command's hash_tab) and merged into as it stands: HashTable.merge(A, B).

    ONE rank (or no process group) is the reference's `--ncores 1`: audfprint.py:473-487 enters `multiproc_add` only for
    ncores > 1; a single process stores straight into `hash_tab` (audfprint.py:177-182) and NOTHING is merged or clipped --
    `counts[k] > depth` stays, as `HashTable.store` leaves it (hash_table.py:120-134), and so do `totalhashes()` and the
    `random.randint(0, count)` draws of a later `add`.  The table is returned untouched (golden `n1` of table_multiproc.npz)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    import numpy as np
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    ht = tb.ht
    nb, depth = 1 << int(ht.hashbits), int(ht.depth)
    meta = dict(names=list(ht.names), hashesperid=np.asarray(ht.hashesperid), depth=depth,
                maxtimebits=int(ht.maxtimebits), hashbits=int(ht.hashbits))
    metas = [None] * world
    dist.all_gather_object(metas, meta)
    if any(m['hashbits'] != meta['hashbits'] or m['maxtimebits'] != meta['maxtimebits'] for m in metas):
        raise ValueError('merge needs tables with the same hashbits / maxtimebits on every rank')
    on_device = dist.get_backend() == 'nccl'
    if on_device and device is None:
        device = torch.device('cuda', torch.cuda.current_device())     # (RCCL needs device tensors on both ends)
    if rank != 0:
        if on_device:
            tp, cp = tb.device_ptrs()
            torch.cuda.synchronize(device)
            dist.send(torch.as_tensor(_DevMem(tp, nb * depth * 4), device=device), dst=0)
            dist.send(torch.as_tensor(_DevMem(cp, nb * 4), device=device), dst=0)
            # an RCCL send returns once it is ENQUEUED; the views above alias the library's own table memory, so the
            # caller must not touch or destroy `tb` before the transfer has drained
            torch.cuda.synchronize(device)
        else:
            tb.finalize()
            dist.send(torch.from_numpy(np.ascontiguousarray(ht.table, dtype=np.uint32).view(np.int32).reshape(-1)), dst=0)
            dist.send(torch.from_numpy(np.ascontiguousarray(ht.counts, dtype=np.int32)), dst=0)
        return None
    novf = []
    if fresh_parent:
        tb.clip_counts()
    if on_device:
        def post(r):
            od = metas[r]['depth']
            bt = torch.empty(nb * od * 4, dtype=torch.uint8, device=device)
            bc = torch.empty(nb * 4, dtype=torch.uint8, device=device)
            return bt, bc, dist.irecv(bt, src=r), dist.irecv(bc, src=r)
        nxt = post(1)
        for r in range(1, world):
            bt, bc, w1, w2 = nxt
            w1.wait()
            w2.wait()
            torch.cuda.synchronize(device)
            if r + 1 < world:
                nxt = post(r + 1)
            m = metas[r]
            novf.append(tb.merge(_RemoteTable(m['names'], m['hashesperid'], m['depth'], m['maxtimebits']),
                                 other_device_ptrs=(bt.data_ptr(), bc.data_ptr())))
    else:
        for r in range(1, world):
            m = metas[r]
            bt = torch.empty(nb * m['depth'], dtype=torch.int32)
            bc = torch.empty(nb, dtype=torch.int32)
            dist.recv(bt, src=r)
            dist.recv(bc, src=r)
            other = _RemoteTable(m['names'], m['hashesperid'], m['depth'], m['maxtimebits'],
                                 table=bt.numpy().view(np.uint32).reshape(nb, m['depth']), counts=bc.numpy())
            novf.append(tb.merge(other))
    return novf
