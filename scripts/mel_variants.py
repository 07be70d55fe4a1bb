"""A/B of fused-filterbank variants (libmispec_NAME.so from scripts/build_variant.py): Mel cfg3, MFCC, Mel 2048/80, more warm-up and
repeats than scripts/fft_variants_time.py; outputs against the shipped library's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import _abi, features  # noqa: E402

dev = "cuda:0"


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


shipped = _abi.load()
libs = [("shipped", shipped)]
for name in os.environ.get("VARIANTS", "u8,u16").split(","):
    path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libmispec_%s.so" % name)
    if name and os.path.exists(path):
        libs.append((name, _abi._load(path, "scripts/build_variant.py")))
libs.append(("shipped again", shipped))
cases = [("Mel cfg3 (1024/512, 128 mels)", features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False), (256, 110250)),
         ("MFCC cfg3", features.MFCC(sr=22050, n_mfcc=20, n_fft=1024, n_mels=128, hop_length=512, verbose=False), (256, 110250)),
         ("Mel 2048/512 80 mels 44.1k", features.MelSpectrogram(sr=44100, n_fft=2048, n_mels=80, hop_length=512, verbose=False), (64, 441000)),
         ("Mel 512/160 40 mels 16k", features.MelSpectrogram(sr=16000, n_fft=512, n_mels=40, hop_length=160, verbose=False), (256, 160000))]
with torch.no_grad():
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
            d = float((y - y0).abs().max() / y0.abs().max())
            row.append("%s %.4f ms%s" % (tag, t, "" if d == 0 else " (max diff %.1e)" % d))
        _abi._lib = shipped
        print("%-34s %s" % (name, " | ".join(row)), flush=True)
