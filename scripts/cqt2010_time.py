"""cfg5 shard (CQT2010v2 / VQT, 96 bins, 64 clips of 30 s) step time per arithmetic.  python scripts/cqt2010_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import features
def timeit(fn, n=40, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(64, 1323000, device="cuda")
for cls, kw in ((features.CQT2010v2, {}), (features.VQT, dict(gamma=0)), (features.VQT, dict(gamma=10))):
    m = cls(sr=44100, hop_length=512, n_bins=96, verbose=False, **kw).cuda()
    ref = None
    for prec in ("fp32", "f16x3", "bf16x3"):
        m.precision = prec
        y = m(x)
        if ref is None: ref = y
        print("%s %s %s: %.4f ms, max |d| / peak vs fp32 %.2e" % (cls.__name__, kw, prec, timeit(lambda: m(x)), float((y - ref).abs().max() / ref.abs().max())), flush=True)
