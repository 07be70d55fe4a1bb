"""Planar contraction timings: the gammatone filterbank (dense, not fusable into the STFT epilogue
for 64 filters x 2048 taps? -- it is a separate launch), the MFCC DCT and the unfused mel path."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnaudio_amd
if os.environ.get("MISPEC_LIB"):  # A/B against another build of the library
    import nnaudio_amd._abi as _abi
    _abi.LIB_PATH = os.environ["MISPEC_LIB"]
from nnaudio_amd import features

def timeit(fn, n=30, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

x = torch.randn(64, 441000, device="cuda")
for prec in ("fp32", "bf16x3"):
    g = features.Gammatonegram(sr=44100, n_fft=2048, n_bins=64, hop_length=512, verbose=False).to("cuda")
    nnaudio_amd.set_precision(prec)
    print("gammatone 64x10s %-7s %.3f ms" % (prec, timeit(lambda: g(x))))
x2 = torch.randn(256, 110250, device="cuda")
for prec in ("fp32", "bf16x3"):
    m = features.MFCC(sr=22050, n_mfcc=20, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to("cuda")
    nnaudio_amd.set_precision(prec)
    print("mfcc 256x5s      %-7s %.3f ms" % (prec, timeit(lambda: m(x2))))
    s = features.STFT(n_fft=1024, hop_length=512, output_format="Complex", iSTFT=True, verbose=False).to("cuda")
    spec = s(x2[:64])
    print("istft 64x5s      %-7s %.3f ms" % (prec, timeit(lambda: s.inverse(spec, onesided=True, length=110250))))
