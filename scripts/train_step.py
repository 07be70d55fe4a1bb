"""STFT(trainable=True) cfg2 batch: zero_grad + forward + mean().backward(), N times (for rocprofv3 --kernel-trace --stats).
    python scripts/train_step.py [precision] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import features  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "auto" else None
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
m = features.STFT(n_fft=2048, hop_length=512, window="hann", output_format="Magnitude", trainable=True, verbose=False).to(dev)
m.precision = prec
x = torch.randn(64, 441000, device=dev)
for _ in range(2):
    m.zero_grad(set_to_none=True)
    m(x).mean().backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    m.zero_grad(set_to_none=True)
    m(x).mean().backward()
e1.record()
torch.cuda.synchronize()
print("train step (%s): %.3f ms" % (prec or "default", e0.elapsed_time(e1) / steps))
