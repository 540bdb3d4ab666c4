#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV (development): for the replayed steps of tools/graphprof.py, how much
of the wall time has >= 1 kernel running, how much is gaps, how much overlap there is, and per kernel the exclusive time
(time during which it was the only kernel running) next to its summed duration.
usage: trace_summary.py <kernel_trace.csv> [skip_fraction | auto]   (auto: everything after the last device pause > 2 ms)"""
import csv
import collections
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"at::native::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]{0,40})?)", n)
    return (m.group(1) if m else n)[:70]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    skip = sys.argv[2] if len(sys.argv) > 2 else "0.3"
    if skip == "auto":
        # the replayed steps follow the last long pause of the device (the graph capture: tens of ms without a kernel)
        cut, end = 0, rows[0][1]
        for i, (s, e, n) in enumerate(rows):
            if s - end > 2_000_000:
                cut = i
            end = max(end, e)
        rows = rows[cut:]
    else:
        rows = rows[int(len(rows) * float(skip)):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    events = []
    for s, e, n in rows:
        events.append((s, 1, n))
        events.append((e, -1, n))
    events.sort()
    active = collections.Counter()
    nact, last = 0, t0
    busy = gap = overlap = 0
    excl = collections.Counter()
    total = collections.Counter()
    calls = collections.Counter()
    for s, e, n in rows:
        total[n] += e - s
        calls[n] += 1
    for t, d, n in events:
        dt = t - last
        if nact == 0:
            gap += dt
        else:
            busy += dt
            if nact == 1:
                excl[next(iter(k for k, v in active.items() if v > 0))] += dt
            else:
                overlap += dt
        last = t
        active[n] += d
        nact += d
    wall = t1 - t0
    print("kernels %d  wall %.3f ms  busy %.3f ms (%.1f%%)  gaps %.3f ms (%.1f%%)  time with >=2 kernels %.3f ms; sum of durations %.3f ms"
          % (len(rows), wall / 1e6, busy / 1e6, 100 * busy / wall, gap / 1e6, 100 * gap / wall, overlap / 1e6, sum(total.values()) / 1e6))
    print("%-72s %8s %10s %10s %8s" % ("kernel", "calls", "sum ms", "excl ms", "avg us"))
    for n, v in total.most_common(45):
        print("%-72s %8d %10.3f %10.3f %8.1f" % (n, calls[n], v / 1e6, excl[n] / 1e6, v / calls[n] / 1e3))
    # where the idle time sits: gaps by the kernel that FOLLOWS them (a kernel that cannot start until something else is done)
    gaps = collections.Counter()
    ngaps = collections.Counter()
    end = rows[0][1]
    for (s, e, n), prev in zip(rows[1:], rows[:-1]):
        if s > end:
            gaps[(prev[2], n)] += s - end
            ngaps[(prev[2], n)] += 1
        end = max(end, e)
    print("largest gaps (after -> before): total us, count, avg us")
    for (a, b), v in gaps.most_common(12):
        print("  %-50s -> %-50s %9.1f %5d %8.1f" % (a[:50], b[:50], v / 1e3, ngaps[(a, b)], v / 1e3 / ngaps[(a, b)]))


if __name__ == "__main__":
    main()
