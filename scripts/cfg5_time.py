"""Step time of the cfg5 shard (64 x 30 s, CQT2010v2) with torch events: production library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import features

m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False).to("cuda:0")
m.precision = os.environ.get("PRECISION") or None
x = torch.randn(64, 1323000, device="cuda:0")
with torch.no_grad():
    for _ in range(10):
        y = m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        y = m(x)
    e1.record()
    torch.cuda.synchronize()
print("cfg5 shard %s: %.4f ms per step" % (m.precision or "f16x3", e0.elapsed_time(e1) / 100))
