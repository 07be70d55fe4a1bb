"""RCCL sanity check on one GPU: a one-rank `nccl` process group, all-reduce, barrier,
nnaudio_amd.dist.sharded_forward, and the ALIASED in-place all_gather_into_tensor (the kernels write
this rank's block into its slice of the gather buffer, which is then both input and output of the
collective) executed for real on RCCL -- with one rank it must leave the buffer bit-identical
(tests/test_gpu_parity.py)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
from nnaudio_amd import dist as D, engine, features
m = features.STFT(n_fft=512, hop_length=128, output_format="Magnitude", verbose=False).to(dev)
x = torch.randn(4, 8000, device=dev)
y = D.sharded_forward(m, x, gather=True)
want = m(x)
# the in-place form, as sharded_forward / ShardedModule issue it at world > 1
full = torch.full_like(want, -1.0)
with engine.output_into(full[0:4]) as slot:
    y2 = m(x)
assert slot.taken and y2.data_ptr() == full.data_ptr()
D.gather_in_place(full, 4, even_alone=True)          # synchronous
w = D.gather_in_place(full, 4, async_op=True, even_alone=True)
for h in w:
    h.wait()
torch.cuda.synchronize()
assert len(w) == 1 and torch.equal(full, want), float((full - want).abs().max())
print("nccl 1-rank ok", tuple(y.shape), float((y - want).abs().max()), "aliased all_gather_into_tensor ok")
dist.destroy_process_group()
