import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnaudio_amd
from nnaudio_amd import features
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
m = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to("cuda")
for B in (1, 16, 64):
    x = torch.randn(B, 441000, device="cuda")
    for prec in ("fp32", "bf16x3"):
        m.precision = prec
        y = m(x)
        print("CQT1992v2 84 bins B=%d x 10 s %-7s %.3f ms" % (B, prec, timeit(lambda: m(x))))
