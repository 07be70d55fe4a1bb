"""Gammatonegram n_fft=2048 hop=512 64 bins, 64 x 10 s @ 44.1 kHz: frame-major route (round 5) against the two kernels
(bins x frames power spectrogram + planar filterbank kernel), and the pieces of each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features  # noqa: E402


def timeit(fn, n=40, w=10):
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


def main():
    x = torch.randn(64, 441000, device="cuda")
    m = features.Gammatonegram(sr=44100, n_fft=2048, n_bins=64, hop_length=512, verbose=False).cuda()
    with torch.no_grad():
        y = m(x)
        t_mod = timeit(lambda: m(x))
        padded = engine.frame_major_filterbank_plan(m, m.gammatone_basis, x, m.stft)
        t_fm = timeit(lambda: m.stft._spectrum(x, engine.EPI_POWER, power=2.0, out_frame_major=padded.shape[1]))
        sp = m.stft._spectrum(x, engine.EPI_POWER, power=2.0, out_frame_major=padded.shape[1])
        t_fb = timeit(lambda: engine.filterbank_frame_major(padded, sp))
        t_std = timeit(lambda: m.stft._spectrum(x, engine.EPI_POWER, power=2.0))
        spec = m.stft._spectrum(x, engine.EPI_POWER, power=2.0)
        t_pl = timeit(lambda: engine.filterbank_autograd(m.gammatone_basis, spec))
        two = engine.filterbank_autograd(m.gammatone_basis, spec)
    print("Gammatonegram 64 x 10 s: module %.4f ms | frame-major power spectrogram %.4f + contraction over the bins %.4f | "
          "before: (bins x frames) spectrogram %.4f + planar filterbank kernel %.4f = %.4f | max diff %.1e of the peak"
          % (t_mod, t_fm, t_fb, t_std, t_pl, t_std + t_pl, float((y - two).abs().max() / two.abs().max())), flush=True)


if __name__ == "__main__":
    main()
