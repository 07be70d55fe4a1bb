"""The cfg5 shard (64 x 30 s) through CQT2010v2 / VQT a few times: the command rocprofv3 wraps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import features

which = sys.argv[1] if len(sys.argv) > 1 else "cqt2010"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cls = features.VQT if which == "vqt" else features.CQT2010v2
m = cls(sr=44100, hop_length=512, n_bins=96, verbose=False).to("cuda:0")
x = torch.randn(64, 1323000, device="cuda:0")
with torch.no_grad():
    for _ in range(n):
        y = m(x)
torch.cuda.synchronize()
print(float(y.abs().max()))
