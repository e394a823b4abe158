#!/usr/bin/env python3
"""Per-forward kernel breakdown from a rocprofv3 --kernel-trace CSV of `bench.py`.
usage: summarize_trace.py <kernel_trace.csv> [forward_index]
The window runs from one corr-build (pf_corr_kernel / pf_corr_rs_kernel) launch pair to the next (one steady-state forward)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pf_corr_kernel" in r["Kernel_Name"] or "pf_corr_rs_kernel" in r["Kernel_Name"]]
starts = idx[0::2]
if len(sys.argv) > 2:
    k = int(sys.argv[2])
else:
    # the shortest complete window = a steady-state HIP-graph replay (bench.py also runs warm-up,
    # eager kernel-profiling and capture forwards, which are longer)
    # (windows of a few kernels are bench.py's back-to-back corr launches for roofline_corr, not forwards)
    spans = [int(rows[starts[j + 1] - 1]["End_Timestamp"]) - int(rows[starts[j]]["Start_Timestamp"]) for j in range(len(starts) - 1)]
    k = min((j for j in range(len(spans)) if starts[j + 1] - starts[j] >= 100), key=lambda j: spans[j])
seg = rows[starts[k]:starts[k + 1]]
agg = collections.OrderedDict()
for r in seg:
    key = r["Kernel_Name"][:100]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, [0, 0.0])
    agg[key][0] += 1
    agg[key][1] += d
tot = sum(v[1] for v in agg.values())
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
# union of busy intervals: time during which at least one kernel is running; the rest is idle gaps
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
union, gaps, cur_s, cur_e = 0, [], iv[0][0], iv[0][1]
where = []                                   # (gap us, offset of its start in the window us, kernel that ends it)
byst = {int(r["Start_Timestamp"]): r["Kernel_Name"] for r in seg}
for a, b in iv[1:]:
    if a > cur_e:
        union += cur_e - cur_s
        gaps.append((a - cur_e) / 1e3)
        where.append(((a - cur_e) / 1e3, (cur_e - iv[0][0]) / 1e3, byst.get(a, "?")))
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
union += cur_e - cur_s
print(f"# forward {k}: kernels {len(seg)}, span {span:.1f} us, sum of kernel durations {tot:.1f} us, "
      f"GPU non-idle {union / 1e3:.1f} us, idle {span - union / 1e3:.1f} us in {len(gaps)} gaps "
      f"(median {sorted(gaps)[len(gaps) // 2] if gaps else 0:.1f} us, max {max(gaps) if gaps else 0:.1f} us)")
for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-102s n=%4d  %9.1f us  avg %8.1f us  %5.1f%%" % (name, v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
if where:
    print("# idle gaps of the window (us, at us from the window's start, next kernel):")
    for g_, at, nxt in sorted(where, reverse=True)[:12]:
        print(f"#   {g_:6.1f}  at {at:8.1f}  before {nxt.replace('(anonymous namespace)::', '')[:80]}")
