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


def pack_host(table, counts, depth):
    """Packed form of a HOST table (include/afp.h): the filled prefix table[k, :min(counts[k], depth)] of every row, bucket
    after bucket -- the only slots HashTable.store / merge write and the only ones a reader looks at (hash_table.py:115-131,
    164, 304-321).  numpy restatement of k_tb_pack_*; stands in for TableBuilder.pack on CPU."""
    import numpy as np
    n = np.minimum(np.maximum(np.asarray(counts, np.int64), 0), int(depth))
    mask = np.arange(int(depth))[None, :] < n[:, None]
    return np.ascontiguousarray(np.asarray(table)[mask], dtype=np.uint32)


def unpack_host(values, counts, depth):
    """The dense (nb, depth) uint32 rows of a packed table; slots outside the filled prefixes are zero."""
    import numpy as np
    n = np.minimum(np.maximum(np.asarray(counts, np.int64), 0), int(depth))
    mask = np.arange(int(depth))[None, :] < n[:, None]
    table = np.zeros((len(n), int(depth)), dtype=np.uint32)
    table[mask] = np.asarray(values, dtype=np.uint32)
    return table


def _lib_error():
    from ._lib import AfpError
    return AfpError


def _wants_device_transport(dist):
    """RCCL ("nccl") moves device memory only: the arrays go GPU to GPU."""
    return dist.get_backend() == 'nccl'


def _alias_probe(tb, device):
    """Can torch (and through it RCCL) take the library's own hipMalloc memory without a copy?  Builds the views the
    device transport would send and touches them; raises if anything on the way does."""
    import torch
    tp, cp = tb.device_ptrs()
    v = torch.as_tensor(_DevMem(cp, 64), device=device)
    if v.data_ptr() != int(cp) or int(v.numel()) != 64:
        raise RuntimeError('torch copied the view of library memory instead of aliasing it')
    int(v[:4].sum().item())
    return True


def merge_tables_to_rank0(tb, dist, device=None, fresh_parent=True, stats=None):
    """The ONE exchange step of the sharded `new -> fpdbase` job (BASELINE configs[3]): every rank has built a private
    table from its clips (audfprint.py:204-224); the parent merges the workers' tables in worker order with
    HashTable.merge (audfprint.py:226-235, hash_table.py:291-323).  Here rank 0 is the parent: ranks 1..N-1 ship their
    tables and rank 0 merges them in rank order (same ids / names order as the reference's loop; over-full buckets draw
    np.random.permutation on rank 0 exactly as there).

    What travels: the PACKED table (include/afp.h) -- counts + the filled prefix of every row, the only slots store / merge
    write and readers read; a 12 500-clip table is 4 + 32 MB that way instead of 424.  (A `tb` without pack() -- a plain
    HashTable behind TableBuilder's interface -- ships its arrays whole, as before.)

    Transport: with RCCL ("nccl") the arrays go GPU to GPU (point-to-point over xGMI) straight out of the sender's library
    memory into a receive buffer that `afp_table_merge_packed_device` reads -- no host round trip; the receive of rank r+1
    is posted before rank r's merge starts.  Every rank first probes locally that torch takes the library's memory without a
    copy (`_alias_probe`) and the ranks agree on the outcome: if ANY rank cannot, all of them stage through torch-owned
    buffers instead (a copy on each side) and a warning is logged -- the first multi-GPU run must not die on first contact.
    A merge that fails on the received device buffers is retried from a host copy of them.  With gloo (CPU tests, or two
    processes sharing one GPU) the arrays are staged through the host.  `tb` is this rank's
    audfprint_amd.table.TableBuilder; returns, on rank 0, the list of over-full bucket counts per merged rank (None
    elsewhere).  `stats` (a dict, optional) receives transport = 'device' | 'staged', bytes_moved (this rank, payload) and
    fallback (why the device transport was not used, or None).

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
    if stats is None:
        stats = {}
    stats.update(transport=None, bytes_moved=0, fallback=None)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    import logging
    import numpy as np
    import torch
    log = logging.getLogger('audfprint_amd.shard')
    rank, world = dist.get_rank(), dist.get_world_size()
    ht = tb.ht
    nb, depth = 1 << int(ht.hashbits), int(ht.depth)
    packed = hasattr(tb, 'pack')
    backend = dist.get_backend()
    want_device = _wants_device_transport(dist)
    if backend == 'nccl' and device is None:
        device = torch.device('cuda', torch.cuda.current_device())     # (RCCL needs device tensors on both ends)
    # ---- agree on the transport BEFORE anything is posted: a send that raises on one side leaves the peer's receive pending
    why = None
    if want_device:
        try:
            _alias_probe(tb, device)
        except Exception as e:       # noqa: BLE001
            why = 'rank %d: %r' % (rank, e)
    nvals = tb.pack() if (packed and rank != 0) else 0
    meta = dict(names=list(ht.names), hashesperid=np.asarray(ht.hashesperid), depth=depth,
                maxtimebits=int(ht.maxtimebits), hashbits=int(ht.hashbits), nvals=int(nvals), packed=bool(packed), why=why)
    metas = [None] * world
    dist.all_gather_object(metas, meta)
    if any(m['hashbits'] != meta['hashbits'] or m['maxtimebits'] != meta['maxtimebits'] for m in metas):
        raise ValueError('merge needs tables with the same hashbits / maxtimebits on every rank')
    whys = [m['why'] for m in metas if m['why']]
    on_device = want_device and not whys
    if want_device and whys:
        stats['fallback'] = '; '.join(whys)
        if rank == 0:
            log.warning('merge_tables_to_rank0: device transport refused (%s) -- staging through torch-owned buffers', stats['fallback'])
    stats['transport'] = 'device' if on_device else 'staged'
    to_dev = (lambda t: t.to(device)) if backend == 'nccl' else (lambda t: t)       # RCCL carries device tensors only

    if rank != 0:
        if on_device:
            if packed:
                vp, cp, n = tb.packed_device_ptrs()
            else:
                (vp, cp), n = tb.device_ptrs(), nb * depth
            torch.cuda.synchronize(device)
            # (int32 views on both ends: the receiver posts int32 buffers, and a send / recv pair should agree on the element type)
            if n:
                dist.send(torch.as_tensor(_DevMem(vp, n * 4), device=device).view(torch.int32), dst=0)
            dist.send(torch.as_tensor(_DevMem(cp, nb * 4), device=device).view(torch.int32), dst=0)
            # an RCCL send returns once it is ENQUEUED; the views above alias the library's own table memory, so the
            # caller must not touch or destroy `tb` before the transfer has drained
            torch.cuda.synchronize(device)
        else:
            if packed:
                vals, cnts = tb.fetch_packed()
            else:
                tb.finalize()
                vals = np.ascontiguousarray(ht.table, dtype=np.uint32).reshape(-1)
                cnts = np.ascontiguousarray(ht.counts, dtype=np.int32)
            n = int(vals.shape[0])
            if n:
                dist.send(to_dev(torch.from_numpy(vals.view(np.int32))), dst=0)
            dist.send(to_dev(torch.from_numpy(cnts)), dst=0)
            if backend == 'nccl':
                torch.cuda.synchronize(device)
        stats['bytes_moved'] = 4 * (n + nb)
        return None

    novf = []
    if fresh_parent:
        tb.clip_counts()
    rdev = device if backend == 'nccl' else None

    def nvals_of(r):
        return metas[r]['nvals'] if metas[r]['packed'] else nb * metas[r]['depth']

    def post(r):
        n = nvals_of(r)
        bv = torch.empty(max(n, 1), dtype=torch.int32, device=rdev)
        bc = torch.empty(nb, dtype=torch.int32, device=rdev)
        if on_device:
            return bv, bc, (dist.irecv(bv[:n], src=r) if n else None), dist.irecv(bc, src=r)
        return bv, bc, None, None

    nxt = post(1)
    for r in range(1, world):
        bv, bc, w1, w2 = nxt
        n, m = nvals_of(r), metas[r]
        if on_device:
            if w1 is not None:
                w1.wait()
            w2.wait()
            torch.cuda.synchronize(device)
        else:
            if n:
                dist.recv(bv[:n], src=r)
            dist.recv(bc, src=r)
            if backend == 'nccl':
                torch.cuda.synchronize(device)
        stats['bytes_moved'] += 4 * (n + nb)
        if r + 1 < world:
            nxt = post(r + 1)

        def host_remote():
            vals = bv[:n].cpu().numpy().view(np.uint32)
            cnts = bc.cpu().numpy()
            table = vals if m['packed'] else vals.reshape(nb, m['depth'])
            return _RemoteTable(m['names'], m['hashesperid'], m['depth'], m['maxtimebits'], table=table, counts=cnts)
        kw = dict(packed=True) if m['packed'] else {}
        if backend == 'nccl':
            try:
                novf.append(tb.merge(_RemoteTable(m['names'], m['hashesperid'], m['depth'], m['maxtimebits']),
                                     other_device_ptrs=(bv.data_ptr(), bc.data_ptr()), **kw))
            except _lib_error() as e:
                # Only a call the library REFUSED before doing anything (bad argument / parameter / state, e.g. device buffers
                # it cannot read) may be retried from a host copy.  A failure after the merge kernel ran -- the overflow fetch,
                # the patch, a runtime error behind the launch -- has already changed the device table and possibly the books:
                # merging the same rank again would duplicate its entries, names and counts.
                if not getattr(e, 'refused', False) or getattr(tb, '_merge_committed', False):
                    raise
                log.warning('merge_tables_to_rank0: merge of rank %d from device buffers failed (%r) -- retrying from a host copy', r, e)
                stats['fallback'] = (stats['fallback'] or '') + ' merge(rank %d): %r' % (r, e)
                novf.append(tb.merge(host_remote(), **kw))
        else:
            novf.append(tb.merge(host_remote(), **kw))
    return novf
