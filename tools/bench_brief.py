#!/usr/bin/env python3
"""Prints the headline of a bench.py JSON line and its top kernels (development helper)."""
import json
import sys
d = json.load(open(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
print(d["value"], "frames/s", d["ms_per_step"], "ms/step", d["config"].get("fp16_overflow_guard"), d["config"].get("launch", "")[:40])
tot = 0.0
for name, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ms"])[:n]:
    print("%-28s %6.1f x %.4f ms = %.3f ms/step  frac %s" % (name, v["launches_per_step"], v["avg_ms"], v["total_ms"] / d["steps"], v.get("frac")))
    for role, r in sorted(v.get("roles", {}).items()):
        print("      /%-20s %6.1f x %.4f ms" % (role, r["launches"] / d["steps"], r["total_ms"] / max(r["launches"], 1)))
print("sum of all hand-written kernels: %.3f ms/step" % (sum(v["total_ms"] for v in d["kernels"].values()) / d["steps"]))
