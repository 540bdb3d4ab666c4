#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE (KiB, rocprofv3 --pmc, one pass each) per launch of every hand-written kernel.
Usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>  ->  JSON on stdout.
Raw counter values are reported as bytes (x1024).  MI355X_MICROARCH.md (section HBM): on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced streaming read.  Calibrated in round 5 on known byte counts (tools/bwprobe.hip reduced by
tools/pmc_calibrate.py, profiles/r05/fetch_size_calibration.txt): the factor is 2.000 for 4-, 8- and 16-byte-per-lane
coalesced reads alike, linear or in the conv producers' 8-plane pattern -- so `fetch_bytes_per_launch` doubles the raw value
for every kernel that reads coalesced rows (all of them except the DCNv2 kernels, whose corner gathers fetch partial lines
and are reported as counted).  `bytes_per_launch` = corrected fetch + write: the `traffic` figure of bench.py."""
import collections
import csv
import glob
import json
import os
import re
import sys


def correction(kernel):
    return 1 if kernel.startswith("dcn_") else 2


def load(d):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(conv_[a-z0-9_]+|fac_[a-z_0-9]+|dcn_[a-z_0-9]+|gn_[a-z_]+|gauss5_[a-z_]+|census_[a-z_]+|src_[a-z_0-9]+|gather_sum_kernel|"
                          r"grad_gather_kernel|pad2d_bwd_kernel|adam_flat_kernel|se_[a-z_]+|ed_[a-z_]+|lap_[a-z_]+|to_c16_kernel|pack_table_[a-z0-9_]+)", name)
            if not m:
                continue
            out[m.group(1)].append(float(r["Counter_Value"]) * 1024.0)
    return out


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fr = sum(f) / len(f) if f else None
        wr = sum(w) / len(w) if w else None
        fc = None if fr is None else fr * correction(k)
        res[k] = {"launches_seen": max(len(f), len(w)),
                  "fetch_bytes_per_launch_raw": None if fr is None else round(fr),
                  "fetch_correction": correction(k),
                  "fetch_bytes_per_launch": None if fc is None else round(fc),
                  "write_bytes_per_launch": None if wr is None else round(wr),
                  "bytes_per_launch": None if fc is None or wr is None else round(fc + wr)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
