"""CQT1992v2 84 bins fp32, 64 clips: does the 32-byte misalignment of successive clips (441 000 samples) cost the frame gather?"""
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


m = features.CQT1992v2(sr=44100, hop_length=512, fmin=32.70, n_bins=84, bins_per_octave=12, verbose=False).cuda()
with torch.no_grad():
    for rnd in range(2):
        for L, note in ((441000, "as benched: clip c starts 32 c bytes off a 128-byte line"), (441024, "128-byte aligned clips"),
                        (441000, "441 000-sample views of 441 024-sample rows (aligned clips)")):
            if "views" in note:
                x = torch.randn(64, 441024, device="cuda")[:, :441000]
            else:
                x = torch.randn(64, L, device="cuda")
            print("round %d  L = %d  %-62s %.4f ms (%d frames)" % (rnd, L, note, timeit(lambda: m(x)), m(x).shape[-1]), flush=True)
