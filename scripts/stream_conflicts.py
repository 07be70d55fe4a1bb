"""LDS conflict counters of the streaming octave kernel per ablation (name the access).
Run under rocprofv3 --pmc (scripts/stream_conflicts.sh): DBG=<bits> python scripts/stream_conflicts.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features

dev = "cuda:0"
m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False).to(dev)
x = torch.randn(int(os.environ.get("B", "64")), 1323000, device=dev)
calls = []
orig = engine.octave_stream


def spy(*a, **k):
    r = orig(*a, **k)
    if r:
        calls.append((a, dict(k)))
    return r


engine.octave_stream = spy
with torch.no_grad():
    m(x)
torch.cuda.synchronize()
engine.octave_stream = orig
dbg = int(os.environ.get("DBG", "0"))
for a, k in calls[:1]:  # the first launch (octaves 0-3): 70 % of the time
    for _ in range(4):
        orig(*a, **dict(k, _debug=dbg))
torch.cuda.synchronize()
