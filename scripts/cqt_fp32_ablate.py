"""CQT1992v2 84 bins fp32 tile kernel: the K loop without its global loads / barriers / MFMAs (benchmarking library)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features  # noqa: E402


def timeit(fn, n=20, w=5):
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


m = features.CQT1992v2(sr=44100, hop_length=512, fmin=32.70, n_bins=84, bins_per_octave=12, verbose=False).to("cuda")
kr, ki = m.cqt_kernels_real, m.cqt_kernels_imag
sup = m._support.get(kr, ki)
sc = torch.sqrt(m.lenghts)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 76
x = torch.randn(B, 441000, device="cuda")
kw = dict(hop=512, pad=m.kernel_width // 2, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, row_scale=sc, row_support=sup, precision="fp32")
with torch.no_grad():
    for name, bits in (("as shipped (16-row tiles)", 0x20000000), ("no global loads in the K loop", 1), ("no MFMAs", 16), ("no barrier", 4),
                       ("32-row tiles", 0x10000000), ("32-row tiles, no global loads", 0x10000001), ("32-row tiles, no MFMAs", 0x10000010)):
        t = timeit(lambda: engine.framed_gemm(x, kr, ki, _debug=bits, **kw))
        print("B=%d %-40s %.4f ms" % (B, name, t), flush=True)
