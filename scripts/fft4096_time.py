"""STFT n_fft = 4096 (64 x 10 s @ 44.1 kHz, hop 1024): the composite FFT-route instance against the contraction kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features  # noqa: E402


def timeit(fn, n=30, w=8):
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


x = torch.randn(64, 441000, device="cuda")
with torch.no_grad():
    for fmt, hop in (("Magnitude", 1024), ("Complex", 1024), ("Magnitude", 512)):
        m = features.STFT(n_fft=4096, hop_length=hop, output_format=fmt, verbose=False).cuda()
        engine.set_fft(True)
        y = m(x)
        t1 = timeit(lambda: m(x))
        engine.set_fft(False)
        r = m(x)
        t0 = timeit(lambda: m(x))
        engine.set_fft(True)
        err = float((y - r).abs().max() / r.abs().max())
        print("STFT 4096/%d %s B=64 x 10 s: fft route %.4f ms, contraction %.4f ms (x%.2f), max diff %.1e of peak, routes differ: %s"
              % (hop, fmt, t1, t0, t0 / t1, err, not torch.equal(y, r)), flush=True)
