"""CQT1992v2 84 bins fp32: time against the number of 128-frame column tiles (is B = 64 -> 431 long workgroups on 256 CUs
a quantization loss?)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import features  # noqa: E402


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


m = features.CQT1992v2(sr=44100, hop_length=512, fmin=32.70, n_bins=84, bins_per_octave=12, verbose=False).to("cuda")
with torch.no_grad():
    for B in (19, 38, 57, 64, 70, 76, 80, 95, 114, 128, 152):
        x = torch.randn(B, 441000, device="cuda")
        t = timeit(lambda: m(x))
        tiles = (B * 862 + 127) // 128
        print("B=%3d: %4d column tiles (%.2f per CU): %.4f ms = %.3f us per tile, %.2f us per clip"
              % (B, tiles, tiles / 256, t, 1e3 * t / tiles, 1e3 * t / B), flush=True)
