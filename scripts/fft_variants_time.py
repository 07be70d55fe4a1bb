"""A/B timings around the FFT route on the GPU box:
  1. n_fft = 256 (zero-extended frames on the 512-point instance; 128 stays on the contraction: 0.9x) against the
     contraction kernels;
  2. the benchmarking build's variant of the n_fft = 1024 instance (8-frame tiles, two workgroups per CU:
     csrc/stft_fft.inl fft_two_per_cu) against the shipped one -- Mel cfg3, MFCC, STFT n_fft = 1024 Magnitude /
     power; outputs must be bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import _abi, engine, features  # noqa: E402

dev = "cuda:0"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.manual_seed(0)
with torch.no_grad():
    # ---- 1. small n_fft
    for n_fft, hop, B, L in ((256, 64, 64, 160000), (128, 32, 64, 80000), (256, 128, 256, 110250)):
        m = features.STFT(n_fft=n_fft, hop_length=hop, output_format="Magnitude", verbose=False).to(dev)
        x = torch.randn(B, L, device=dev)
        y = m(x)
        t_fft = timeit(lambda: m(x))
        engine.set_fft(False)
        g = m(x)
        t_gemm = timeit(lambda: m(x))
        engine.set_fft(True)
        err = float((y - g).abs().max() / g.abs().max())
        print("STFT n_fft=%d hop=%d B=%d L=%d: fft %.4f ms, contraction %.4f ms (x%.2f), max diff %.1e of peak, routes differ: %s"
              % (n_fft, hop, B, L, t_fft, t_gemm, t_gemm / t_fft, err, not torch.equal(y, g)), flush=True)
        del x, y, g

    # ---- 2. variants of the tile geometry (scripts/build_variant.py NAME ...) vs the shipped library
    shipped = _abi.load()
    libs = [("shipped", shipped)]
    for name in os.environ.get("VARIANTS", "old,v4").split(","):
        path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libmispec_%s.so" % name)
        if name and os.path.exists(path):
            libs.append((name, _abi._load(path, "scripts/build_variant.py")))
    libs.append(("shipped again", shipped))
    cases = [
        ("Mel cfg3 (1024/512, 128 mels)", features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False), (256, 110250)),
        ("MFCC cfg3", features.MFCC(sr=22050, n_mfcc=20, n_fft=1024, n_mels=128, hop_length=512, verbose=False), (256, 110250)),
        ("STFT 1024/256 Magnitude", features.STFT(n_fft=1024, hop_length=256, output_format="Magnitude", verbose=False), (64, 441000)),
        ("STFT 1024/256 Complex", features.STFT(n_fft=1024, hop_length=256, output_format="Complex", verbose=False), (64, 441000)),
        ("STFT 1024/512 Magnitude", features.STFT(n_fft=1024, hop_length=512, output_format="Magnitude", verbose=False), (64, 441000)),
        ("Mel 1024/256 229 mels", features.MelSpectrogram(sr=44100, n_fft=1024, n_mels=229, hop_length=256, verbose=False), (64, 441000)),
        ("STFT 512/128 Magnitude", features.STFT(n_fft=512, hop_length=128, output_format="Magnitude", verbose=False), (64, 441000)),
        ("STFT 512/128 Complex", features.STFT(n_fft=512, hop_length=128, output_format="Complex", verbose=False), (64, 441000)),
        ("Mel 512/160 40 mels 16k", features.MelSpectrogram(sr=16000, n_fft=512, n_mels=40, hop_length=160, verbose=False), (256, 160000)),
        ("STFT 256/64 Magnitude", features.STFT(n_fft=256, hop_length=64, output_format="Magnitude", verbose=False), (64, 160000)),
        ("STFT 2048/512 Magnitude", features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False), (64, 441000)),
    ]
    for name, m, shape in cases:
        m = m.to(dev)
        x = torch.randn(*shape, device=dev)
        row, y0 = [], None
        for tag, lib in libs:
            _abi._lib = lib
            y = m(x).clone()
            t = timeit(lambda: m(x))
            if y0 is None:
                y0 = y
            row.append("%s %.4f ms%s" % (tag, t, "" if y is y0 else " (max diff %.1e of peak)" % float((y - y0).abs().max() / y0.abs().max())))
        _abi._lib = shipped
        print("%-30s %s" % (name, " | ".join(row)), flush=True)
        del x, y0, y
