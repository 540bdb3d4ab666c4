#!/usr/bin/env python3
"""FETCH_SIZE calibration on a known byte count (MI355X_MICROARCH.md, HBM section: 'calibrate on a known byte count in your own
access pattern'): reduces the counter CSV of
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -o p -- tools/bin/bwprobe 512
to counted / actual bytes per variant of tools/bwprobe.hip (linear 1 KB runs vs the conv producers' 8 planes x 128 B pattern).
usage: pmc_calibrate.py OUT_DIR actual_megabytes"""
import collections
import csv
import glob
import re
import sys

out, mb = sys.argv[1], float(sys.argv[2])
rows = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        m = re.search(r"probe<(\d+), (\d+)>", r["Kernel_Name"])
        if m:
            rows[("planes 16 B/lane" if int(m.group(2)) else "linear 16 B/lane", int(m.group(1)),
                  int(r.get("Workgroup_Size", 0)))].append(float(r["Counter_Value"]))
        m = re.search(r"probe_narrow<(\d+), (\d+)>", r["Kernel_Name"])
        if m:
            rows[("linear %2d B/lane" % int(m.group(2)), int(m.group(1)), int(r.get("Workgroup_Size", 0)))].append(float(r["Counter_Value"]))
for (kind, nl, wg), v in sorted(rows.items()):
    kb = sum(v) / len(v)          # FETCH_SIZE is reported in kilobytes
    print("%s  loads in flight %2d  workgroup %4d : counted %8.1f MB of %6.0f MB actual  -> factor %.3f  (%d dispatches)"
          % (kind, nl, wg, kb / 1024.0, mb, mb / (kb / 1024.0), len(v)))
