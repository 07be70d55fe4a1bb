#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats CSV + counter CSVs) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("# rocprofv3 summary for", root)
for f in find("*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  %-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
            r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"),
            r.get("AverageNs"), r.get("Percentage")))
for f in find("*counter_collection.csv"):
    print("\n## counters (%s)" % os.path.relpath(f, root))
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:80]
        agg[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0) or 0)
        cnt[(k, r.get("Counter_Name"))] += 1
    for k, d in agg.items():
        if not any(t in k for t in ("framed_", "split_signal", "fold", "octave_pyramid", "octave_stream", "fir_decimate", "clip_absmax", "stft_fft", "cqt_chain")):
            continue
        print("  kernel:", k)
        for c, v in sorted(d.items()):
            n = cnt[(k, c)]
            print("     %-34s sum=%.6g  per-dispatch=%.6g  (dispatch rows=%d)" % (c, v, v / max(n, 1), n))


# derived: effective clock and MFMA busy fraction of the dominant kernel
try:
    dom, dur, best = None, None, -1.0
    for f in find("*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if any(t in r.get("Name", "") for t in ("framed_", "octave_pyramid", "octave_stream", "stft_fft", "cqt_chain")) and float(r["TotalDurationNs"]) > best:
                best = float(r["TotalDurationNs"])
                dom, dur = r["Name"], float(r["AverageNs"]) * 1e-9
    vals = {}
    for f in find("*counter_collection.csv"):
        agg = defaultdict(float); n = defaultdict(int)
        for r in csv.DictReader(open(f)):
            if dom and r.get("Kernel_Name", "") == dom:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in agg:
            vals[k] = agg[k] / n[k]
    if dur and "GRBM_GUI_ACTIVE" in vals:
        cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
        print("\n## derived (dominant kernel, per dispatch): %s" % dom[:100])
        print("  avg duration (kernel-trace pass)  : %.3f ms" % (dur * 1e3))
        print("  GRBM_GUI_ACTIVE/8 (cycles)        : %.4g  -> effective clock %.2f GHz" % (cyc, cyc / dur / 1e9))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            print("  MFMA busy fraction                : %.1f %%  (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD * cycles))" % (100 * vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))
        if "SQ_WAVE_CYCLES" in vals:
            w = vals["SQ_WAVE_CYCLES"]
            for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in vals:
                    print("  %-34s: %.1f %% of wave cycles" % (k, 100 * vals[k] / w))
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in vals or "SQ_INSTS_VALU_MFMA_MOPS_F32" in vals:
            for k in ("SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
                if k in vals:
                    print("  %-34s: %.4g per dispatch" % (k, vals[k]))
        if "FETCH_SIZE" in vals:
            print("  FETCH_SIZE (KB) per dispatch      : %.4g  (x2 on gfx950 for 16B/lane streams -> %.1f MB read)" % (vals["FETCH_SIZE"], 2 * vals["FETCH_SIZE"] * 1024 / 1e6))
        if "WRITE_SIZE" in vals:
            print("  WRITE_SIZE (KB) per dispatch      : %.4g  (%.1f MB written)" % (vals["WRITE_SIZE"], vals["WRITE_SIZE"] * 1024 / 1e6))
except Exception as e:
    print("derived metrics failed:", e)
