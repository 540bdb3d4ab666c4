#!/usr/bin/env python3
"""PyTorch-native kernels (at::native, runtime copies / fills) of the LAST replayed training step in a rocprofv3 kernel trace of
tools/graphprof.py, grouped by kernel and launch size (development tool: what is left outside the library).
usage: gluetrace.py <kernel_trace.csv> [launches per step = 563]"""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 563
rows = rows[-per_step:]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "at::native" in n or "rocclr" in n:
        short = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n)
        m = re.match(r"([A-Za-z_0-9]+)", short)
        tag = re.search(r"(direct_copy|CUDAFunctor\w*add|FillFunctor|MulFunctor|Abs|where|compare|leaky\w*|sigmoid\w*|Mean|sum_functor|MaxNan)", short)
        key = m.group(1) + (" " + tag.group(1) if tag else "")
        threads = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        d[(key, threads)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print("%-58s threads=%9d n=%3d sum %7.1f us" % (k[:58], g, len(v), sum(v)))
print("total %.1f us of a %.1f us step (%.1f %%)" % (tot, span, 100 * tot / span))
