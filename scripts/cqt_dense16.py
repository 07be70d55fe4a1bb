"""CQT1992v2 (CQT84 workload: 84 bins, 64 clips of 10 s at 44.1 kHz) in f16x3 on the staged dense kernel
(natural tap order) vs the strip kernel (hop-periodic order) vs fp32 / bf16x3: step time and the largest
difference from the fp32 tile kernels relative to the peak.  python scripts/cqt_dense16.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import features
def timeit(fn, n=50, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(64, 441000, device="cuda")
for fmt in ("Magnitude", "Complex"):
    m = features.CQT1992v2(sr=44100, hop_length=512, fmin=32.70, n_bins=84, bins_per_octave=12,
                           output_format=fmt, verbose=False).to("cuda")
    m.precision = "fp32"
    ref = m(x)
    for prec, strip in (("fp32", False), ("bf16x3", False), ("f16x3", False), ("f16x3", True)):
        m.precision, m.hop_periodic = prec, strip
        y = m(x)
        err = float((y - ref).abs().max() / ref.abs().max())
        print("%s %s%s: %.4f ms, max |d| / peak vs fp32 %.2e" % (fmt, prec, " strip" if strip else "", timeit(lambda: m(x)), err), flush=True)
