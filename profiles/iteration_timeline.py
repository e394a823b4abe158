#!/usr/bin/env python3
"""Timeline of ONE refinement iteration from a rocprofv3 --kernel-trace CSV of bench.py: every kernel between two
consecutive pf_motion_prep launches of the steady-state forward, with start / end relative to the first and the queue.
usage: iteration_timeline.py <kernel_trace.csv> [iteration_index_from_the_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pf_motion_prep" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a, b = marks[-k - 1], marks[-k]
t0 = int(rows[a]["Start_Timestamp"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_114pf_elem_kernelI", "elem<")[:70]
print(f"# iteration span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} kernels")
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>3}  {short(r['Kernel_Name'])}")
