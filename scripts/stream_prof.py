"""Phase clock and ablations of the streaming octave kernel on the cfg5 shard (benchmarking build).

  NO_STAMPS=1   ablation timings only (the phase clock costs workgroup 7 about 0.1 us per stamp)
  DBG_STAMP=0,1 debug bits of the launches the phase clock is read from (one report per entry)
  PRECISION=bf16x3  the module's arithmetic (default: f16x3)
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features

dev = "cuda:0"
m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False).to(dev)
m.precision = os.environ.get("PRECISION") or None  # (None: the module default, f16x3)
B = int(os.environ.get("B", "64"))
x = torch.randn(B, 1323000, device=dev)
N_SLOTS = 12
stamps = torch.zeros(8 * 32 * N_SLOTS, dtype=torch.int64, device=dev)
NO_STAMPS = os.environ.get("NO_STAMPS", "0") == "1"
if not NO_STAMPS:
    os.environ["MISPEC_OS_STAMPS"] = str(stamps.data_ptr())
DBG_STAMP = [int(v) for v in os.environ.get("DBG_STAMP", "0").split(",")]

calls = []
orig = engine.octave_stream


def spy(*a, **k):
    r = orig(*a, **k)
    if r:  # (a refused launch falls back to the pyramid kernel: nothing to time)
        calls.append((a, dict(k)))
    else:
        print("refused:", len(a[1]), "levels, headroom", k.get("fir_headroom_bits"))
    return r


engine.octave_stream = spy
with torch.no_grad():
    y = m(x)
torch.cuda.synchronize()
engine.octave_stream = orig
print("launches:", len(calls), [len(c[0][1]) for c in calls])


def timeit(fn, n=200):
    for _ in range(100):  # (long enough for the clocks to settle: the first timing of a process is otherwise the slowest)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ABLATIONS = [
    (0, "full"), (1, "no FIR MFMAs"), (2, "no bank tiles"), (4, "no global stores"), (8, "no DMA"),
    (3, "no FIR, no banks"), (15, "nothing"), (16, "bank waves prio 0 (not 3)"), (32, "FIR waves prio 0 (not 2)"), (48, "all waves prio 0"),
    (1 | 4, "no FIR: tiles without stores"), (1 | 64, "no FIR: tiles without MFMAs"),
    (1 | 128, "no FIR: tile fragments from one address"),
    (1 | 4 | 64 | 128, "no FIR: tiles without address math, MFMAs, stores"),
    (1 | 2 | 256, "no FIR, no tiles: ingest without split + writes"), (1 | 2 | 512, "no FIR, no tiles: no scale check"),
    (1 | 2 | 8, "no FIR, no tiles: no DMA"), (1 | 2 | 8 | 256 | 512, "no FIR, no tiles: staging read + barrier only"),
]
if os.environ.get("ONLY"):  # e.g. ONLY=0,16,32,48
    keep = {int(v) for v in os.environ["ONLY"].split(",")}
    ABLATIONS = [a for a in ABLATIONS if a[0] in keep]
NAMES_FIR = {1: "check", 2: "setup+frag0", 3: "mfma loop", 4: "epilogue + wait loads", 5: "staging read + dma issue", 10: "max+split+write", 11: "barrier"}
NAMES_BANK = {1: "check", 5: "mode 2: fetch both quarters", 4: "mode 2: commit both", 6: "edge tiles (mode 2, no tile: wait for the requests)",
              7: "plain: loads + MFMAs", 8: "plain: epilogue + stores", 9: "-", 10: "ingest (mode 0)", 11: "barrier"}


def phase_report(li, a, k, dbg):
    stamps.zero_()
    orig(*a, **dict(k, _debug=dbg))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(8, 32, N_SLOTS).astype(np.float64) * 0.01  # us
    for w in range(8):
        d = s[w]
        ok = d[:, 0] > 0
        if ok.sum() < 8:
            continue
        names = NAMES_FIR if w < 4 else NAMES_BANK
        per = np.median(np.diff(d[ok, 0]))
        rows = d[ok][6:]  # steady state
        kinds = ("tile",) if w < 4 else ("tile", "plain")
        for kind in kinds:
            if w < 4:
                sel = rows[:, 10] > 0
            else:
                sel = rows[:, 8] > 0 if kind == "tile" else rows[:, 8] == 0
            r = rows[sel]
            if not len(r):
                continue
            out, prev = [], 0
            for slot in names:
                if np.all(r[:, slot] == 0):
                    continue
                out.append("%s %.2f" % (names[slot], np.median(r[:, slot] - r[:, prev])))
                prev = slot
            print("launch %d dbg %d wave %d (%s steps, period %.2f us): %s" % (li, dbg, w, kind, per, " | ".join(out)), flush=True)


for li, (a, k) in enumerate(calls):
    if os.environ.get("SKIP_ABLATIONS", "0") != "1":
        for dbg, name in ABLATIONS:
            t = timeit(lambda: orig(*a, **dict(k, _debug=dbg)))
            print("launch %d  %-52s %.4f ms" % (li, name, t), flush=True)
        for nseg in (2, 4, 8, 16):
            t = timeit(lambda: orig(*a, **dict(k, _debug=0, n_segments=nseg)))
            print("launch %d  n_segments=%d  %.4f ms" % (li, nseg, t), flush=True)
    if not NO_STAMPS:
        for dbg in DBG_STAMP:
            phase_report(li, a, k, dbg)
