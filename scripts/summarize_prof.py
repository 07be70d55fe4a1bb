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
        if "framed_gemm" not in k:
            continue
        print("  kernel:", k)
        for c, v in sorted(d.items()):
            n = cnt[(k, c)]
            print("     %-34s sum=%.6g  per-dispatch=%.6g  (dispatch rows=%d)" % (c, v, v / max(n, 1), n))
