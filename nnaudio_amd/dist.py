"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed with
the "nccl" backend = RCCL over xGMI).

A spectrogram of one waveform depends on nothing else in the batch, and the bases are a few
MB, so the bases are replicated and the batch is cut into contiguous blocks: rank r owns
clips [r*B/G, (r+1)*B/G).  No collective is needed to *compute*; the only exchange is the
optional reassembly of the full (B, bins, frames) tensor.  ``sharded_forward`` allocates that
tensor once, lets the kernels of the module's final launch write this rank's block straight
into its slice (``engine.output_into``) and all-gathers IN PLACE -- no staging copy on either
side; with ``chunks > 1`` the local block is computed in sub-blocks whose gathers run
asynchronously under the next sub-block's kernels.
This is what the reference's only multi-GPU usage (nn.DataParallel: scatter the batch,
replicate the module, gather the outputs; tests/test_cqt.py:273-291) amounts to.
"""
import torch
import torch.distributed as dist

from . import engine


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


def _aliasing_ok(group):
    # RCCL all-gathers in place when the input is this rank's slice of the output; gloo (CPU
    # tests) is given a private copy of the input instead
    return dist.get_backend(group) == "nccl"


def gather_in_place(full, n_clips, group=None, async_op=False, even_alone=False):
    """All-gather ``full`` (``(n_clips, ...)``, contiguous), whose block ``shard_bounds(rank)``
    this rank has already filled, so that every rank holds every block.  Equal blocks: ONE
    in-place ``all_gather_into_tensor``.  Ragged split (``n_clips % world != 0``): one in-place
    broadcast per rank (no padding, no staging).  Returns the list of async work handles
    (empty when ``async_op`` is False).  A single rank has nothing to gather and returns at once,
    unless ``even_alone`` asks for the collective anyway (the aliased in-place form on RCCL with
    one rank: tests/test_gpu_parity.py)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not even_alone:
        return []
    works = []
    if n_clips % world == 0:
        lo, hi = shard_bounds(n_clips, world, rank)
        mine = full[lo:hi]
        if not _aliasing_ok(group):
            mine = mine.clone()
        w = dist.all_gather_into_tensor(full, mine, group=group, async_op=async_op)
        if async_op:
            works.append(w)
        return works
    for src in range(world):
        lo, hi = shard_bounds(n_clips, world, src)
        if hi > lo:
            w = dist.broadcast(full[lo:hi], src=dist.get_global_rank(group, src) if group else src,
                               group=group, async_op=async_op)
            if async_op:
                works.append(w)
    return works


def gather_batch(y_local, n_clips, group=None):
    """All-gather per-rank blocks ``(b_r, ...)`` that were NOT computed in place into the full
    ``(n_clips, ...)`` tensor (one copy of the local block, then ``gather_in_place``)."""
    world = dist.get_world_size(group)
    if world == 1:
        return y_local
    rank = dist.get_rank(group)
    full = torch.empty((n_clips,) + tuple(y_local.shape[1:]), dtype=y_local.dtype,
                       device=y_local.device)
    lo, hi = shard_bounds(n_clips, world, rank)
    full[lo:hi].copy_(y_local)
    gather_in_place(full, n_clips, group=group)
    return full


def sharded_forward(module, x_full, gather=True, group=None, chunks=1, **fwd):
    """Run ``module`` on this rank's block of ``x_full`` (the same tensor on every rank) and,
    with ``gather``, reassemble the batch on every rank.

    The full output is allocated once; the module's final launch writes this rank's clips into
    their slice of it (no local output tensor, no copy) and the all-gather runs in place.  With
    ``chunks > 1`` the local block is processed in ``chunks`` sub-blocks and the gather of
    sub-block j is issued asynchronously (RCCL's own stream) while sub-block j+1 computes; this
    needs equal blocks (``n_clips % world == 0`` and the local block divisible by ``chunks``),
    otherwise one gather is issued at the end."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = x_full.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    x_local = x_full[lo:hi]
    if not gather:
        return module(x_local, **fwd)
    if world == 1:
        return module(x_local, **fwd)
    per = hi - lo
    even = n % world == 0 and chunks > 1 and per % chunks == 0
    if not even:
        chunks = 1
    sub = per // chunks if per else 0
    full = None
    works = []
    for j in range(chunks):
        a, b = (lo + j * sub, lo + (j + 1) * sub) if chunks > 1 else (lo, hi)
        xs = x_full[a:b]
        if full is None:
            # the output shape is only known after the first forward: run it, then allocate the
            # gather buffer and move the block (once per process lifetime would need the shape in
            # advance; callers that know it pass `out_like` through engine.output_into themselves)
            y = module(xs, **fwd)
            full = torch.empty((n,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
            full[a:b].copy_(y)
            del y
        else:
            with engine.output_into(full[a:b]) as slot:
                y = module(xs, **fwd)
            if not slot.taken or y.data_ptr() != full[a:b].data_ptr():
                full[a:b].copy_(y)  # a module whose last allocation is not its output
            del y
        if chunks > 1:
            # gather sub-block j of every rank while the next one computes
            outs = [full[r * per + j * sub: r * per + (j + 1) * sub] for r in range(world)]
            mine = outs[rank] if _aliasing_ok(group) else outs[rank].clone()
            works.append(dist.all_gather(outs, mine, group=group, async_op=True))
    if chunks == 1:
        gather_in_place(full, n, group=group)
    for w in works:
        w.wait()
    return full


class ShardedModule(torch.nn.Module):
    """``module`` wrapped so that ``forward(x_full)`` is ``sharded_forward`` with a persistent
    gather buffer: from the second call on (same input shape) the kernels write straight into
    the buffer's slice and nothing is allocated or copied.

    The returned tensor IS that buffer: the next ``forward`` overwrites it (results kept in a list,
    or handed to a consumer on another stream, must be copied -- or construct with ``clone=True``,
    which returns a private copy per call at the price of one device copy)."""

    def __init__(self, module, group=None, gather=True, clone=False):
        super().__init__()
        self.module = module
        self.group = group
        self.gather = gather
        self.clone = clone
        self._full = None

    def forward(self, x_full, **fwd):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        n = x_full.shape[0]
        lo, hi = shard_bounds(n, world, rank)
        if not self.gather or world == 1:
            return self.module(x_full[lo:hi], **fwd)
        full = self._full
        if full is not None and (full.shape[0] != n or full.device != x_full.device):
            full = None
        y = None
        if full is not None:
            with engine.output_into(full[lo:hi]) as slot:
                y = self.module(x_full[lo:hi], **fwd)
            if tuple(y.shape[1:]) != tuple(full.shape[1:]) or y.dtype != full.dtype:
                full = None  # output format changed (the slot was not taken): new buffer for this y
            elif not slot.taken or y.data_ptr() != full[lo:hi].data_ptr():
                full[lo:hi].copy_(y)
        if full is None:
            if y is None:
                # the output's shape past the batch dimension from ONE clip (not from a whole forward that is
                # then copied): the block itself is computed straight into its slice of the new buffer
                with torch.no_grad():
                    probe = self.module(x_full[lo:lo + 1], **fwd) if hi > lo else self.module(x_full[:1], **fwd)
                full = torch.empty((n,) + tuple(probe.shape[1:]), dtype=probe.dtype, device=probe.device)
                del probe
                with engine.output_into(full[lo:hi]) as slot:
                    y = self.module(x_full[lo:hi], **fwd)
                if not slot.taken or y.data_ptr() != full[lo:hi].data_ptr():
                    full[lo:hi].copy_(y)
            else:
                full = torch.empty((n,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
                full[lo:hi].copy_(y)
            self._full = full
        gather_in_place(full, n, group=self.group)
        return full.clone() if self.clone else full
