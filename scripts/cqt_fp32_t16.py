"""CQT1992v2 84 bins, fp32 (the default module's arithmetic): 16-row support tiles (round 5) against the 32-row ones
(benchmarking library, bit 0x10000000)."""
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
for B in (1, 8, 64):
    x = torch.randn(B, 441000, device="cuda")
    kw = dict(hop=512, pad=m.kernel_width // 2, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, row_scale=sc, row_support=sup, precision="fp32")
    with torch.no_grad():
        y_mod = m(x)
        y16 = engine.framed_gemm(x, kr, ki, **kw)
        y32 = engine.framed_gemm(x, kr, ki, _debug=0x10000000, **kw)
        ref = engine.framed_gemm(x, kr, ki, reference_kernel=True, **kw) if B == 1 else None
        t_mod = timeit(lambda: m(x))
        t16 = timeit(lambda: engine.framed_gemm(x, kr, ki, **kw))
        t32 = timeit(lambda: engine.framed_gemm(x, kr, ki, _debug=0x10000000, **kw))
    peak = float(y32.abs().max())
    print("B=%3d: module %.4f ms | 16-row tiles %.4f ms, 32-row tiles %.4f ms (x%.3f) | module == 16-row call: %s | "
          "16 vs 32: %.2e of the peak%s"
          % (B, t_mod, t16, t32, t32 / t16, torch.equal(y_mod, y16), float((y16 - y32).abs().max()) / peak,
             "" if ref is None else " | vs reference kernel: 16-row %.2e, 32-row %.2e"
             % (float((y16 - ref).abs().max()) / peak, float((y32 - ref).abs().max()) / peak)), flush=True)
