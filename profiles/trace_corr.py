#!/usr/bin/env python3
"""Print the kernel durations of the corr kernels from a rocprofv3 --kernel-trace CSV, in launch order,
averaged over consecutive groups of `reps` launches (the groups of profiles/ab_corr.py).
usage: trace_corr.py <kernel_trace.csv> <reps>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pf_corr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
reps = int(sys.argv[2])
grp = reps + 1                     # ab_corr.py: one untimed launch + reps timed ones per variant
for i in range(0, len(rows), grp):
    seg = rows[i:i + grp]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg]
    name = "ring" if "ring" in seg[0]["Kernel_Name"] else "tile"
    print(f"{i // grp:3d} {name}: n={len(d)} avg {sum(d) / len(d):7.1f} us  min {min(d):7.1f}  max {max(d):7.1f}  "
          f"LDS {seg[0].get('LDS_Block_Size', '?')} VGPR {seg[0].get('VGPR_Count', '?')} grid {seg[0].get('Grid_Size', '?')} wg {seg[0].get('Workgroup_Size', '?')}")
