"""PyTorch's own kernels (at::native / rocclr copies) in ONE training step of profiles/time_train_step.py under rocprofv3
--kernel-trace: per (kernel, grid size) the launch count and the time, largest first -- what is left to replace.
    python3 profiles/native_census.py <t_kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
step = rows[ends[-2] + 1:ends[-1] + 1]
by = {}
tot = 0
for r in step:
    n = r["Kernel_Name"]
    if "pf_" in n and "at::native" not in n:
        continue
    m = re.search(r"(\w+Functor\w*(<[^>]*>)?|direct_copy\w*|copyBuffer|fillBuffer\w*|CatArray\w*|index\w*|reduce_kernel|clamp\w*|addcmul\w*)", n)
    key = ((m.group(1) if m else n[:60]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
    d = by.setdefault(key, [0, 0])
    d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    d[1] += 1
    tot += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"# {sum(c for _, c in by.values())} PyTorch / runtime kernels in the step, {tot / 1e3:.1f} us in all (of {len(step)} launches)")
for (k, g), (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / 1e3:9.1f} us  x{c:4d}  grid {g:10d}  {k[:80]}")
