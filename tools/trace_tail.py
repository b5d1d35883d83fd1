"""Keeps the last 1500 dispatches of a rocprofv3 kernel trace as a text timeline (start, idle gap in front, duration, grid,
kernel) next to it and deletes the full trace (too large to travel back from the GPU box).  usage: trace_tail.py <dir>"""
import csv
import glob
import os
import sys

out = sys.argv[1]
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-1500:]
    t0, prev = int(rows[0]["Start_Timestamp"]), int(rows[0]["Start_Timestamp"])
    with open(os.path.join(out, "trace_tail.txt"), "w") as g:
        for r in rows:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            g.write("%10.1f gap %7.1f dur %8.1f grid %8s wg %5s %s\n" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")),
                                                                        r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"][:90]))
            prev = e
    os.remove(f)
