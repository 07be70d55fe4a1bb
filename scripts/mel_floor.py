"""Where Mel cfg3's time goes: the 1024-point transform alone in its three output forms on cfg3's batch (256 x 5 s @ 22.05 kHz,
hop 512): Magnitude (bins x frames tile + flush), Power written frame-major (no tile), and the fused mel reduction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features  # noqa: E402


def timeit(fn, n=200, w=50):
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


x = torch.randn(256, 110250, device="cuda")
st = features.STFT(n_fft=1024, hop_length=512, output_format="Magnitude", verbose=False).cuda()
mel = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False).cuda()
with torch.no_grad():
    t_mag = timeit(lambda: st(x))
    t_pow = timeit(lambda: st._spectrum(x, engine.EPI_POWER, power=2.0))
    t_fm = timeit(lambda: st._spectrum(x, engine.EPI_POWER, power=2.0, out_frame_major=544))
    t_mel = timeit(lambda: mel(x))
print("cfg3 batch, n_fft 1024 hop 512: STFT Magnitude %.4f ms | Power %.4f | Power frame-major %.4f | MelSpectrogram (fused reduction) %.4f"
      % (t_mag, t_pow, t_fm, t_mel), flush=True)
