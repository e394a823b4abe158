#!/usr/bin/env python3
"""Timeline of ONE graph replay from a rocprofv3 kernel trace: every kernel of forward number `which` with start / end relative
to the forward's first kernel, its queue and duration -- used to read the encoder phase (everything before the first lookup).
   python profiles/encoder_timeline.py <t_kernel_trace.csv> [which] [until_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
until = float(sys.argv[3]) if len(sys.argv) > 3 else 3500.0
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda t: t[0])
# a forward starts with its input stage (one pf_prepare_images launch per forward; pf_normalise_images before round 6's fusion)
starts = [i for i, k in enumerate(ks) if "pf_prepare_images" in k[2] or "pf_normalise_images" in k[2]]
which = min(which, len(starts) - 2)
i0 = starts[which]
i1 = starts[which + 1] if which + 1 < len(starts) else len(ks)
t0 = ks[i0][0]
qmap = {}
for s, e, n, q in ks[i0:i1]:
    if (s - t0) / 1e3 > until:
        break
    qi = qmap.setdefault(q, len(qmap))
    short = n.replace("(anonymous namespace)::", "").replace("pfconv::", "")
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{qi}  {short[:110]}")
