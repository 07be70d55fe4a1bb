import sys, torch
sys.path.insert(0, "/root/repo")
import nnaudio_amd
from nnaudio_amd import features
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x3 = torch.randn(256, 110250, device="cuda")
x2 = torch.randn(64, 441000, device="cuda")
mods = [("Mel cfg3", features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False), x3),
        ("MFCC cfg3 shape", features.MFCC(sr=22050, n_mfcc=20, n_fft=1024, n_mels=128, hop_length=512, verbose=False), x3),
        ("Gammatone n_fft 2048, 64 bins, B=64 x 10 s", features.Gammatonegram(sr=44100, n_fft=2048, n_bins=64, hop_length=512, verbose=False), x2),
        ("STFT cfg2 Complex", features.STFT(n_fft=2048, hop_length=512, output_format="Complex", verbose=False), x2)]
for name, m, x in mods:
    m = m.to("cuda")
    for prec in ("fp32", "bf16x3"):
        m.precision = prec
        if hasattr(m, "stft"): m.stft.precision = prec
        if hasattr(m, "melspec_layer"): m.melspec_layer.precision = prec; m.melspec_layer.stft.precision = prec
        nnaudio_amd.set_precision(prec)
        print("%-45s %-7s %.3f ms" % (name, prec, timeit(lambda: m(x))))
