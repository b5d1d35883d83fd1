#!/usr/bin/env python3
"""Timeline of one bench step from a `rocprofv3 --kernel-trace` csv: every launch behind the sketch GEMMs with its start
offset, duration and the idle gap in front of it (host work, read-backs, launch latency), and the totals per kernel.

    python tools/trace_gaps.py gpurun_out/<run>/kt/kt_kernel_trace.csv [--step -1] [--all]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--step", type=int, default=-1, help="which step of the run (default: the last)")
    ap.add_argument("--all", action="store_true", help="print every launch, not only the summary")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
    rows.sort()
    # a step starts at a randn launch followed by the big sketch GEMMs; split at the random fill of the sketching matrix
    starts = [i for i, r in enumerate(rows) if r[2].startswith("randn_kernel")]
    # the rhs fill of the bench is one randn before the steps: keep those followed by a dgemm within 4 launches
    starts = [i for i in starts if any(rows[j][2].startswith("dgemm_kernel") for j in range(i + 1, min(i + 5, len(rows))))]
    if not starts:
        raise SystemExit("no step found")
    s = starts[a.step]
    e = starts[starts.index(s) + 1] if starts.index(s) + 1 < len(starts) else len(rows)
    step = rows[s:e]
    t0 = step[0][0]
    # the tail starts after the last big sketch launch (the longest kernels of the step)
    big = max(range(len(step)), key=lambda i: step[i][1] - step[i][0])
    last_big = max(i for i in range(len(step)) if step[i][1] - step[i][0] > 0.2 * (step[big][1] - step[big][0]))
    tail = step[last_big + 1:]
    print("step: %d launches, %.3f ms; sketch part %.3f ms; tail: %d launches, %.3f ms" % (
        len(step), (step[-1][1] - t0) / 1e6, (step[last_big][1] - t0) / 1e6, len(tail), (tail[-1][1] - step[last_big][1]) / 1e6))
    prev = step[last_big][1]
    busy = 0
    gaps = []
    per = defaultdict(lambda: [0, 0.0, 0.0])
    for st, en, nm, wg in tail:
        gap = max(0, st - prev)
        dur = en - st
        busy += dur
        gaps.append((gap, nm))
        p = per[nm]
        p[0] += 1; p[1] += dur / 1e3; p[2] += gap / 1e3
        if a.all:
            print("%9.1f us  +%7.1f gap  %8.1f us  wg %6d  %s" % ((st - step[last_big][1]) / 1e3, gap / 1e3, dur / 1e3, wg, nm))
        prev = max(prev, en)
    tot = (tail[-1][1] - step[last_big][1]) / 1e3
    print("tail: kernels busy %.1f us, idle %.1f us (%.0f %%)" % (busy / 1e3, tot - busy / 1e3, 100 * (1 - busy / 1e3 / tot)))
    big_gaps = sorted(gaps, reverse=True)[:12]
    print("largest gaps (us, in front of):", ", ".join("%.0f %s" % (g / 1e3, n.split("<")[0]) for g, n in big_gaps))
    print("%-46s %5s %10s %10s" % ("kernel", "n", "busy us", "gap us"))
    for nm, (n, d, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print("%-46s %5d %10.1f %10.1f" % (nm[:46], n, d, g))


if __name__ == "__main__":
    main()
