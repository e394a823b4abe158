"""Largest idle gaps (no kernel running on any queue) inside the shortest forward window of a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# forward windows start at the image-rotate kernel
starts = [i for i, k in enumerate(K) if "PfImgRot" in k[2]]
wins = [(K[starts[j + 1] - 1][1] - K[starts[j]][0], starts[j], starts[j + 1]) for j in range(len(starts) - 1)]
span, a, b = min(wins)
W = K[a:b]
print("window kernels", len(W), "span us", span / 1e3)
gaps, end = [], W[0][1]
prev = W[0]
for k in W[1:]:
    if k[0] > end: gaps.append((k[0] - end, prev[2][:60], k[2][:60], (end - W[0][0]) / 1e3))
    if k[1] > end: end, prev = k[1], k
for g in sorted(gaps, reverse=True)[:12]:
    print(f"{g[0] / 1e3:7.1f} us at t={g[3]:8.1f} us   after {g[1]}   before {g[2]}")
