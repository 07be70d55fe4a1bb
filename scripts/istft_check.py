"""Inverse STFT: frames from the inverse FFT vs from the contraction with the synthesis basis -- parity and
time (cfg2-sized spectrogram, round trip).  python scripts/istft_check.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rng = np.random.default_rng(0)
for n_fft, hop, B, L in ((2048, 512, 3, 40000), (1024, 256, 2, 9000), (512, 128, 4, 5001), (2048, 1024, 1, 2100)):
    m = features.STFT(n_fft=n_fft, hop_length=hop, iSTFT=True, output_format="Complex", verbose=False).cuda()
    x = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).cuda()
    X = m(x)
    engine.set_fft(True); y = m.inverse(X, length=L)
    engine.set_fft(False); r = m.inverse(X, length=L)
    err = float((y - r).abs().max() / r.abs().max())
    rt = float((y - x).abs().max())
    print("n_fft %d hop %d: inverse fft vs contraction %.2e of the peak (equal %s), round trip max |d| %.2e" % (n_fft, hop, err, bool(torch.equal(y, r)), rt), flush=True)
    assert err < 2e-6 and not torch.equal(y, r)
    im = features.iSTFT(n_fft=n_fft, hop_length=hop, verbose=False).cuda()
    engine.set_fft(True); y2 = im(X, onesided=True, length=L)
    assert float((y2.float() - r).abs().max() / r.abs().max()) < 2e-6
m = features.STFT(n_fft=2048, hop_length=512, iSTFT=True, output_format="Complex", verbose=False).cuda()
x = torch.randn(64, 441000, device="cuda")
engine.set_fft(True); X = m(x)
for on, fused in ((True, True), (True, False), (False, False)):
    engine.set_fft(on)
    engine.set_istft_fused(fused)
    print("cfg2-sized inverse, fft=%s fused=%s: %.4f ms" % (on, fused, timeit(lambda: m.inverse(X, length=441000))), flush=True)
