import os, sys
import torch
sys.path.insert(0, "/root/repo")
from nnaudio_amd import _abi, engine, features
dev = "cuda:0"
def timeit(fn, n=200):
    for _ in range(100): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shipped = _abi.load()
hx = _abi._load(os.path.join(os.path.dirname(_abi.LIB_PATH), "libmispec_hx.so"), "variant")
for name, m, shape in [("STFT 2048/512 Magnitude cfg2", features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False), (64, 441000)),
                       ("STFT 2048/512 power, short clips", features.STFT(n_fft=2048, hop_length=300, output_format="Magnitude", verbose=False), (7, 5000))]:
    m = m.to(dev); x = torch.randn(*shape, device=dev)
    with torch.no_grad():
        row = []; y0 = None
        for tag, lib in (("shipped", shipped), ("hx", hx), ("shipped again", shipped)):
            _abi._lib = lib
            y = m(x).clone(); t = timeit(lambda: m(x))
            if y0 is None: y0 = y
            row.append("%s %.4f ms (max diff %.1e of peak)" % (tag, t, float((y - y0).abs().max() / y0.abs().max())))
        _abi._lib = shipped
        print(name, " | ".join(row), flush=True)
