"""STFT through the FFT path vs the contraction kernels: parity on random clips (all output formats, the
three n_fft, edge frames, odd hops, freq_bins) and step time at cfg2.  python scripts/fft_check.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
def timeit(fn, n=50, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rng = np.random.default_rng(0)
for n_fft, hop, B, L, kw in ((2048, 512, 3, 30000, {}), (1024, 256, 2, 9001, {}), (512, 128, 5, 4000, dict(pad_mode="constant")),
                             (2048, 333, 2, 20000, dict(freq_bins=700)), (512, 64, 2, 3000, dict(center=False)),
                             (1024, 512, 1, 600, {}), (2048, 2048, 2, 50000, dict(window="hamming"))):
    x = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).cuda()
    for fmt in ("Complex", "Magnitude", "Phase"):
        m = features.STFT(n_fft=n_fft, hop_length=hop, output_format=fmt, verbose=False, **kw).cuda()
        m.precision = "fp32"
        engine.set_fft(True); y = m(x)
        engine.set_fft(False); r = m(x)
        assert y.shape == r.shape
        if fmt == "Phase":
            mm = features.STFT(n_fft=n_fft, hop_length=hop, output_format="Magnitude", verbose=False, **kw).cuda()(x)
            strong = mm > 1e-2 * mm.max()
            d = (torch.exp(1j * y) - torch.exp(1j * r)).abs()[strong].max()
            print("n_fft %d hop %d %s %s: max phase distance (strong bins) %.2e" % (n_fft, hop, kw, fmt, float(d)))
            assert d < 1e-3
        else:
            err = float((y - r).abs().max() / r.abs().max())
            print("n_fft %d hop %d %s %s: fft vs fp32 contraction %.2e of the peak, equal %s" % (n_fft, hop, kw, fmt, err, bool(torch.equal(y, r))))
            assert err < 2e-6 and not torch.equal(y, r)
x = torch.randn(64, 441000, device="cuda")
for fmt in ("Magnitude", "Complex"):
    m = features.STFT(n_fft=2048, hop_length=512, output_format=fmt, verbose=False).cuda()
    for on in (True, False):
        engine.set_fft(on)
        print("cfg2 %s fft=%s: %.4f ms" % (fmt, on, timeit(lambda: m(x))), flush=True)
# phases of the kernel (benchmarking build): 0x1 no stores, 0x2 no post-processing / epilogue, 0x4 no passes
if "--ablate" in sys.argv:
    m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).cuda()
    engine.set_fft(True)
    prep = engine.prepare_basis(m.wcos, m.wsin, "fp32", hop=512)
    A = 0x10000000
    for bits in (0, 0x80, 0x40, 0x48, 0x41, 8, 1, 2, 4, 6, 7):
        fn = lambda: engine.framed_gemm(x, m.wcos, m.wsin, hop=512, pad=1024, pad_mode=2, epilogue=engine.EPI_MAGNITUDE,
                                        precision="fp32", _debug=A | bits, **prep)
        print("ablation bits %x: %.4f ms" % (bits, timeit(fn)), flush=True)
# MelSpectrogram / MFCC: the filterbank reduced in the FFT kernel's tile vs in the contraction's epilogue
if "--mel" in sys.argv:
    xm = torch.randn(256, 110250, device="cuda")
    for cls, kw in ((features.MelSpectrogram, dict(sr=22050, n_fft=1024, n_mels=128, hop_length=512)),
                    (features.MelSpectrogram, dict(sr=22050, n_fft=2048, n_mels=80, hop_length=256, power=1.0)),
                    (features.MFCC, dict(sr=22050, n_mfcc=20, n_fft=1024, n_mels=128, hop_length=512)),
                    (features.Gammatonegram, dict(sr=22050, n_fft=1024, n_bins=64, hop_length=512))):
        m = cls(verbose=False, **kw).cuda()
        engine.set_fft(True); y = m(xm); t1 = timeit(lambda: m(xm))
        engine.set_fft(False); r = m(xm); t0 = timeit(lambda: m(xm))
        err = float((y - r).abs().max() / r.abs().max())
        print("%s %s: fft %.4f ms, contraction %.4f ms, max |d| / peak %.2e" % (cls.__name__, kw, t1, t0, err), flush=True)
        assert err < 1e-5 and not torch.equal(y, r)
