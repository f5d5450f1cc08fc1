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
