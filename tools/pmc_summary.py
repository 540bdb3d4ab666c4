#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE (KiB, rocprofv3 --pmc, one pass each) per launch of every hand-written kernel.
Usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>  ->  JSON on stdout.
Raw counter values are reported as bytes (x1024).  MI355X_MICROARCH.md (section HBM): on gfx950 FETCH_SIZE reports half of
the bytes of a 16-B-per-lane streaming read -- `fetch_bytes_per_launch` doubles the raw value for the kernels whose staging
loads are 16-byte quads (QUAD below); dword-per-lane staging is outside the guide's calibration and left as counted.
`bytes_per_launch` = corrected fetch + write: the `traffic` figure of bench.py."""
import collections
import csv
import glob
import json
import os
import re
import sys


QUAD = ("conv_fwd_bf16x3_ws", "conv_fwd_f16_ws", "conv_wgrad_f16_tr", "fac_fwd_tile_f32", "fac_bwd_rows_f32")


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
        fr = sum(f) / len(f) if f else None
        wr = sum(w) / len(w) if w else None
        fc = None if fr is None else fr * (2 if k in QUAD else 1)
        res[k] = {"launches_seen": max(len(f), len(w)),
                  "fetch_bytes_per_launch_raw": None if fr is None else round(fr),
                  "fetch_correction": 2 if k in QUAD else 1,
                  "fetch_bytes_per_launch": None if fc is None else round(fc),
                  "write_bytes_per_launch": None if wr is None else round(wr),
                  "bytes_per_launch": None if fc is None or wr is None else round(fc + wr)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
