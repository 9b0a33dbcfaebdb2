#!/usr/bin/env python3
"""Steady-state view of a rocprofv3 kernel trace: per kernel, the average duration over all calls (what
`--stats` prints) next to the average without the first `skip` calls (the untimed warm-up launches of bench.py,
whose first one also pays the code-object load and cold caches).  bench.py's HIP-event figure covers only the
timed launches, so it is the second column that has to agree with `roofline.avg_launch_ms`.

usage: kt_steady.py <kernel_trace.csv> [skip=warm-up launches] [name filter] [group=launches per timed repeat]

With `group` the steady launches are cut into consecutive groups of that many (bench.py times K launches per repeat, with a
barrier and a counter read-back between repeats) and span_per_call is the mean over groups of (first start .. last end) /
group: the figure bench.py reports as roofline.avg_launch_ms from HIP events.
"""
import csv
import sys
from collections import OrderedDict


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    filt = sys.argv[3] if len(sys.argv) > 3 else "k_wavestack"
    group = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    calls = OrderedDict()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if filt not in name:
                continue
            calls.setdefault(name, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    print("kernel,calls,avg_all_ms,skip,avg_steady_ms,min_ms,max_steady_ms,first_ms,span_per_call_steady_ms")
    for name, c in calls.items():
        c.sort()
        d = [(e - s) / 1e6 for s, e in c]
        st = d[skip:] if len(d) > skip else d
        cs = c[skip:] if len(c) > skip else c
        # wall clock from the first steady launch's start to the last one's end, per launch: with the frame
        # pipeline consecutive launches overlap, so this is shorter than the average duration
        if group and len(cs) >= group:
            gs = [cs[k:k + group] for k in range(0, len(cs) - group + 1, group)]
            span = sum((g[-1][1] - g[0][0]) / 1e6 / len(g) for g in gs) / len(gs)
        else:
            span = (cs[-1][1] - cs[0][0]) / 1e6 / len(cs)
        print('"%s",%d,%.4f,%d,%.4f,%.4f,%.4f,%.4f,%.4f' % (name, len(d), sum(d) / len(d), skip, sum(st) / len(st),
                                                          min(d), max(st), d[0], span))


if __name__ == "__main__":
    main()
