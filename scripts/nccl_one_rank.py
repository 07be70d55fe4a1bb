"""RCCL sanity check on one GPU: a one-rank `nccl` process group, all-reduce, barrier and
nnaudio_amd.dist.sharded_forward with the all-gather reassembly (tests/test_gpu_parity.py)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
from nnaudio_amd import dist as D, features
m = features.STFT(n_fft=512, hop_length=128, output_format="Magnitude", verbose=False).to(dev)
x = torch.randn(4, 8000, device=dev)
y = D.sharded_forward(m, x, gather=True)
print("nccl 1-rank ok", tuple(y.shape), float((y - m(x)).abs().max()))
dist.destroy_process_group()
