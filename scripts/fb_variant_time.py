"""A/B of the dense filterbank launch's tile configuration: python scripts/build_variant.py NAME DEFINE, then
VARIANTS=NAME python scripts/fb_variant_time.py on the GPU box (Gammatonegram of cfg2's batch; two-call MFCC)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import _abi, engine, features

dev = "cuda:0"


def timeit(fn, n=200):
    for _ in range(100):
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
for name in os.environ.get("VARIANTS", "fbn128").split(","):
    path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libmispec_%s.so" % name)
    if name and os.path.exists(path):
        libs.append((name, _abi._load(path, "scripts/build_variant.py")))
libs.append(("shipped again", shipped))
engine.set_mfcc_fused(False)
cases = [
    ("Gammatonegram 2048/512 64 bins", features.Gammatonegram(sr=44100, n_fft=2048, n_bins=64, hop_length=512, verbose=False), (64, 441000)),
    ("Gammatonegram 1024/256 64 bins", features.Gammatonegram(sr=22050, n_fft=1024, n_bins=64, hop_length=256, verbose=False), (64, 220500)),
    ("Gammatonegram 2048/512, 8 clips", features.Gammatonegram(sr=44100, n_fft=2048, n_bins=64, hop_length=512, verbose=False), (8, 441000)),
]
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
            row.append("%s %.4f ms%s" % (tag, t, "" if y is y0 else " (max diff %.1e of peak)" % float((y - y0).abs().max() / y0.abs().max())))
        _abi._lib = shipped
        print("%-32s %s" % (name, " | ".join(row)), flush=True)
