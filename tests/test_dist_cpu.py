"""world_size-2 gloo test of the batch-sharding / reassembly logic (no GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Fake(torch.nn.Module):
    """Stands in for a feature module: a per-clip function of the waveform."""

    def forward(self, x):
        return torch.stack((x[:, ::7].cumsum(-1), x[:, ::7] ** 2), 1)


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nnaudio_amd import dist as D

        x = torch.arange(n_clips * 70, dtype=torch.float32).reshape(n_clips, 70)
        full = _Fake()(x)
        y = D.sharded_forward(_Fake(), x, gather=True)
        ok = torch.equal(y, full)
        lo, hi = D.shard_bounds(n_clips, world, rank)
        loc = D.sharded_forward(_Fake(), x, gather=False)
        ok = ok and torch.equal(loc, full[lo:hi])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [8, 5])
def test_sharded_forward_gloo(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(2))
    assert got == {0: True, 1: True}


def test_shard_bounds_cover_batch():
    from nnaudio_amd.dist import shard_bounds

    for n in (1, 7, 64, 128, 513):
        for g in (1, 2, 4, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
