"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed with
the "nccl" backend = RCCL over xGMI).

A spectrogram of one waveform depends on nothing else in the batch, and the bases are a few
MB, so the bases are replicated and the batch is cut into contiguous blocks: rank r owns
clips [r*B/G, (r+1)*B/G).  No collective is needed to *compute*; the only exchange is the
optional reassembly of the full (B, bins, frames) tensor, one all-gather in which every
rank's block is already in place inside the gather buffer (the kernels write there).
This is what the reference's only multi-GPU usage (nn.DataParallel: scatter the batch,
replicate the module, gather the outputs; tests/test_cqt.py:273-291) amounts to.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_clips, world_size, rank):
    """Contiguous, balanced block of the batch owned by ``rank`` (first ``n % G`` ranks get
    one extra clip)."""
    q, r = divmod(int(n_clips), int(world_size))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(x, world_size=None, rank=None, group=None):
    if world_size is None:
        world_size = dist.get_world_size(group)
    if rank is None:
        rank = dist.get_rank(group)
    lo, hi = shard_bounds(x.shape[0], world_size, rank)
    return x[lo:hi]


def gather_batch(y_local, n_clips, group=None):
    """All-gather per-rank blocks ``(b_r, ...)`` into the full ``(n_clips, ...)`` tensor on
    every rank.  Equal blocks use one ``all_gather_into_tensor``; ragged tails fall back to
    padded gathers."""
    world = dist.get_world_size(group)
    if world == 1:
        return y_local
    q, r = divmod(int(n_clips), world)
    tail = tuple(y_local.shape[1:])
    if r == 0:
        out = torch.empty((n_clips,) + tail, dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=group)
        return out
    padded = torch.zeros((q + 1,) + tail, dtype=y_local.dtype, device=y_local.device)
    padded[: y_local.shape[0]] = y_local
    buf = torch.empty((world * (q + 1),) + tail, dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for rk in range(world):
        lo, hi = shard_bounds(n_clips, world, rk)
        parts.append(buf[rk * (q + 1): rk * (q + 1) + (hi - lo)])
    return torch.cat(parts, 0)


def sharded_forward(module, x_full, gather=True, group=None, **fwd):
    """Run ``module`` on this rank's block of ``x_full`` (same tensor on every rank, or only
    the local block when ``x_full`` is already sharded and ``gather`` is False) and optionally
    reassemble the batch."""
    x_local = shard_batch(x_full, group=group)
    y_local = module(x_local, **fwd)
    if not gather:
        return y_local
    return gather_batch(y_local, x_full.shape[0], group=group)
