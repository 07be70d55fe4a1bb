import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
DEV = "cuda:0"
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 441000, device=DEV)
c = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to(DEV)
r1 = c(x); r2 = c(x); r3 = c(x)
print("fp32 repeatable:", bool(torch.equal(r1, r2) and torch.equal(r2, r3)))
c.precision = "bf16x3"
ys = [c(x) for _ in range(4)]
for i in range(1, 4):
    d = (ys[i] - ys[0]).abs()
    print("bf16 run", i, "vs 0: differing", int((d > 0).sum()), "max", float(d.max()))
pk = r1.abs().max()
for i in range(4):
    d = (ys[i] - r1).abs()
    print("bf16 run", i, "vs fp32: max err/peak %.3e" % float(d.max() / pk), "bad", int((d > 1e-4 * pk).sum()))
