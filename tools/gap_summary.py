#!/usr/bin/env python3
"""Where the idle gaps of a rocprofv3 --kernel-trace timeline are (development): gap histogram and the largest gaps with the
kernels on either side.  usage: gap_summary.py <kernel_trace.csv> [skip_fraction]"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_summary import short  # noqa: E402

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = rows[int(len(rows) * skip):]
gaps = []
end = rows[0][1]
prev = rows[0][2]
for s, e, n in rows[1:]:
    if s > end:
        gaps.append((s - end, prev, n))
    if e > end:
        end, prev = e, n
tot = sum(g[0] for g in gaps)
print("gaps %d, total %.3f ms" % (len(gaps), tot / 1e6))
for lo, hi in [(0, 1e3), (1e3, 3e3), (3e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e12)]:
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print("  %8.0f-%-8.0f ns: %6d gaps %9.3f ms" % (lo, hi, len(sel), sum(sel) / 1e6))
pairs = collections.Counter()
for d, a, b in gaps:
    pairs[(a[:40], b[:40])] += d
print("largest summed gaps by (previous kernel -> next kernel):")
for (a, b), d in pairs.most_common(25):
    print("  %9.3f ms  %-40s -> %s" % (d / 1e6, a, b))
