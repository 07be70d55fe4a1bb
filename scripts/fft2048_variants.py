"""A/B of the n_fft = 2048 FFT kernel's variants (scripts/build_variant.py NAME MISPEC_FFT2048_MODE=n ...) on the GPU box:
cfg2 Magnitude, same process, same input; every variant's output against the shipped library's (must be identical)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import _abi, features  # noqa: E402

dev = "cuda:0"


def timeit(fn, n=100):
    for _ in range(20):
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
shipped = _abi.load()
libs = [("shipped", shipped)]
for name in os.environ.get("VARIANTS", "m0,m1,m2").split(","):
    path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libmispec_%s.so" % name)
    if name and os.path.exists(path):
        libs.append((name, _abi._load(path, "scripts/build_variant.py")))
libs.append(("shipped again", shipped))
cases = [("STFT 2048/512 Magnitude B=64 (cfg2)", dict(n_fft=2048, hop_length=512, output_format="Magnitude"), (64, 441000)),
         ("STFT 2048/512 Magnitude B=7 x 3 s", dict(n_fft=2048, hop_length=512, output_format="Magnitude"), (7, 132301)),
         ("STFT 2048/256 Magnitude B=32", dict(n_fft=2048, hop_length=256, output_format="Magnitude"), (32, 441000)),
         ("STFT 2048/512 power=2 (Mel-free Power epilogue) B=64", None, (64, 441000))]
with torch.no_grad():
    for title, kw, shape in cases:
        if kw is None:
            continue
        m = features.STFT(verbose=False, **kw).to(dev)
        x = torch.randn(*shape, device=dev)
        y0, row = None, []
        for tag, lib in libs:
            _abi._lib = lib
            y = m(x).clone()
            t = timeit(lambda: m(x))
            if y0 is None:
                y0 = y
            d = float((y - y0).abs().max() / y0.abs().max())
            row.append("%s %.4f ms%s" % (tag, t, "" if d == 0 else " (max diff %.1e of peak)" % d))
        _abi._lib = shipped
        print("%-50s %s" % (title, " | ".join(row)), flush=True)
