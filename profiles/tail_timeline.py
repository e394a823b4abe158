#!/usr/bin/env python3
"""Timeline of the END of one replayed forward from a rocprofv3 --kernel-trace CSV of bench.py: every kernel from the last
pf_motion_prep launch of a steady-state forward (= the start of the last refinement iteration, in which branch B's update is dead
work and the mask head runs) to the upsampling kernel, with start / end relative to the first and the queue.
usage: tail_timeline.py <kernel_trace.csv> [forward_index_from_the_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ups = [i for i, r in enumerate(rows) if "pf_upsample" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
end = ups[-k]
a = max(i for i, r in enumerate(rows[:end]) if "pf_motion_prep" in r["Kernel_Name"])
t0 = int(rows[a]["Start_Timestamp"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_114pf_elem_kernelI", "elem<")[:70]  # noqa: E731
print(f"# last iteration + mask head + upsampling: {(int(rows[end]['End_Timestamp']) - t0) / 1e3:.1f} us, {end - a + 1} kernels")
for r in rows[a:end + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>3}  {short(r['Kernel_Name'])}")
