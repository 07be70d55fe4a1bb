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


class _FakeInPlace(torch.nn.Module):
    """The same function, allocating its output the way the feature modules' last launch does
    (engine.alloc_out): lets the sharding layer place it inside the gather buffer."""

    def __init__(self):
        super().__init__()
        self.placed = 0
        self.batches = []

    def forward(self, x):
        from nnaudio_amd import engine

        self.batches.append(x.shape[0])
        ref = _Fake()(x)
        out = engine.alloc_out(ref.shape, x.device)
        out.copy_(ref)
        return out


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nnaudio_amd import dist as D

        x = torch.arange(n_clips * 70, dtype=torch.float32).reshape(n_clips, 70)
        full = _Fake()(x)
        ok = True
        for mod in (_Fake(), _FakeInPlace()):
            y = D.sharded_forward(mod, x, gather=True)
            ok = ok and torch.equal(y, full)
            y = D.sharded_forward(mod, x, gather=True, chunks=2)  # async sub-block gathers
            ok = ok and torch.equal(y, full)
        lo, hi = D.shard_bounds(n_clips, world, rank)
        loc = D.sharded_forward(_Fake(), x, gather=False)
        ok = ok and torch.equal(loc, full[lo:hi])
        # persistent gather buffer: from the second call on the module writes into its slice
        sm = D.ShardedModule(_FakeInPlace())
        y1 = sm(x)
        ok = ok and torch.equal(y1, full)
        y2 = sm(2 * x)
        ok = ok and torch.equal(y2, _Fake()(2 * x)) and y2.data_ptr() == y1.data_ptr()
        # the first call learns the output shape from ONE clip, then computes its block in place once
        ok = ok and sm.module.batches == [1, hi - lo, hi - lo]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_clips", [(2, 8), (2, 5), (3, 7), (3, 12)])
def test_sharded_forward_gloo(world, n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(world))
    assert got == {r: True for r in range(world)}


def test_output_into_is_one_shot_and_shape_checked():
    from nnaudio_amd import engine

    buf = torch.empty(4, 3, 5)
    with engine.output_into(buf) as slot:
        a = engine.alloc_out((4, 3, 6), buf.device)       # wrong shape: fresh memory
        b = engine.alloc_out((4, 3, 5), buf.device, zero=True)
        c = engine.alloc_out((4, 3, 5), buf.device)       # already taken
    assert slot.taken and b.data_ptr() == buf.data_ptr() and float(b.abs().sum()) == 0.0
    assert a.data_ptr() != buf.data_ptr() and c.data_ptr() != buf.data_ptr()
    d = engine.alloc_out((4, 3, 5), buf.device)           # outside the context: never placed
    assert d.data_ptr() != buf.data_ptr()
    with pytest.raises(RuntimeError):
        engine.output_into(buf[:, :, ::2])


def test_shard_bounds_cover_batch():
    from nnaudio_amd.dist import shard_bounds

    for n in (1, 7, 64, 128, 513):
        for g in (1, 2, 4, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
