#!/usr/bin/env python3
"""Per-kernel durations of the last repetition of profiles/enc_alone.py from its rocprofv3 kernel trace."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fill" in r["Kernel_Name"].lower()]
marks = marks[-3:]
for name, a, b in (("fnet", marks[0], marks[1]), ("cnet", marks[1], marks[2])):
    seg = rows[a + 1:b]
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    print(f"== {name}: {len(seg)} kernels, sum {tot:.1f} us, span {span:.1f} us")
    for r in seg:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pfconv::", "").replace("void ", "")
        g = r.get("Grid_Size", "?"); w = r.get("Workgroup_Size", "?")
        print(f"   {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {g:>9} wg {w:>4}  {n[:90]}")
