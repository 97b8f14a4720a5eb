"""Multi-GPU sharding of a batch: chunks are independent, so chunk i belongs to rank
i mod world — no data-path collective.  torch.distributed is used only for the control plane
(barrier, max-over-ranks timing, gathering per-rank byte counts)."""


def shard_indices(n_chunks, rank, world):
    """indices of the chunks rank owns: i with i mod world == rank"""
    return range(rank, n_chunks, world)


def shard_count(n_chunks, rank, world):
    return len(shard_indices(n_chunks, rank, world))


def aggregate(dist, device, seconds, unc_bytes):
    """whole-job view: (max time over ranks, total uncompressed bytes).  dist may be None for world 1."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds, unc_bytes
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    b = torch.tensor([float(unc_bytes)], dtype=torch.float64, device=device)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(t.item()), float(b.item())
