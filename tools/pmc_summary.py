#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE (KiB, rocprofv3 --pmc, one pass each) per launch of every hand-written kernel.
Usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>  ->  JSON on stdout.
Raw counter values are reported as bytes (x1024); the x2 correction of MI355X_MICROARCH.md applies to 16-B-per-lane
streaming reads only (the FAC kernels), the dword-per-lane conv staging is uncalibrated and left as counted."""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(conv_[a-z0-9_]+|fac_[a-z_0-9]+|dcn_[a-z_0-9]+|gn_[a-z_]+|gauss5_[a-z_]+|census_[a-z_]+|src_[a-z_]+|gather_sum_kernel)", name)
            if not m:
                continue
            out[m.group(1)].append(float(r["Counter_Value"]) * 1024.0)
    return out


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        res[k] = {"launches_seen": max(len(f), len(w)),
                  "fetch_bytes_per_launch_raw": round(sum(f) / len(f)) if f else None,
                  "write_bytes_per_launch": round(sum(w) / len(w)) if w else None}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
