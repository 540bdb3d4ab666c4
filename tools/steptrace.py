#!/usr/bin/env python3
"""Per-(kernel, grid) duration statistics of a rocprofv3 kernel trace (development tool).
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py ...; python3 tools/steptrace.py DIR [substr ...]"""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
want = sys.argv[2:] or [""]
d = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    n = r["Kernel_Name"]
    if not any(w in n for w in want):
        continue
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)(<[^(]*>)?\(", n.replace("(anonymous namespace)::", ""))
    short = ((m.group(1) + (m.group(2) or "")) if m else n)[:44]
    key = "%-44s grid=%7s,%-3s,%-2s wg=%s" % (short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    tot += sum(v)
    print("%s n=%4d med %8.1f min %8.1f max %8.1f sum %9.1f us" % (k, len(v), v[len(v) // 2], v[0], v[-1], sum(v)))
print("total %.1f us" % tot)
