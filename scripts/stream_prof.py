"""Phase clock and ablations of the streaming octave kernel on the cfg5 shard (benchmarking build)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
from nnaudio_amd.features import _cqt_common as C

dev = "cuda:0"
m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False).to(dev)
B = int(os.environ.get("B", "64"))
x = torch.randn(B, 1323000, device=dev)
stamps = torch.zeros(8 * 32 * 4, dtype=torch.int64, device=dev)
os.environ["MISPEC_OS_STAMPS"] = str(stamps.data_ptr())

calls = []
orig = engine.octave_stream

def spy(*a, **k):
    calls.append((a, dict(k)))
    return orig(*a, **k)

engine.octave_stream = spy
with torch.no_grad():
    y = m(x)
torch.cuda.synchronize()
engine.octave_stream = orig
print("launches:", len(calls), [len(c[0][1]) for c in calls])

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for li, (a, k) in enumerate(calls):
    for dbg, name in ((0, "full"), (1, "no FIR MFMAs"), (2, "no bank tiles"), (4, "no global stores"), (8, "no DMA"),
                      (3, "no FIR, no banks"), (15, "nothing")):
        t = timeit(lambda: orig(*a, **dict(k, _debug=dbg)))
        print("launch %d  %-18s %.4f ms" % (li, name, t), flush=True)
    for nseg in (2, 4, 8, 16):
        t = timeit(lambda: orig(*a, **dict(k, _debug=0, n_segments=nseg)))
        print("launch %d  n_segments=%d  %.4f ms" % (li, nseg, t), flush=True)
    stamps.zero_()
    orig(*a, **dict(k, _debug=0))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(8, 32, 4).astype(np.float64) * 0.01  # us
    for w in (0, 3, 4, 7):
        d = s[w]
        ok = d[:, 0] > 0
        if ok.sum() < 3:
            continue
        rows = d[ok]
        top = (rows[:, 1] - rows[:, 0])
        work = (rows[:, 2] - rows[:, 1])
        bot = (rows[:, 3] - rows[:, 2])
        per = np.diff(rows[:, 0])
        print("launch %d wave %d: steps %d  step period %.2f us (median)  top %.2f  work %.2f  bottom(wait+barrier) %.2f"
              % (li, w, ok.sum(), np.median(per), np.median(top), np.median(work), np.median(bot)))
        print("     first steps work:", np.round(work[:12], 2))
