"""List the kernels that overlap pf_flow_out_* in a rocprofv3 kernel trace (csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
K.sort()
fo = [k for k in K if "pf_flow_out" in k[2]]
for s, e, n, q, st in fo[-3:]:
    print(f"{n[:40]} [{e - s} ns] queue {q} stream {st}")
    for s2, e2, n2, q2, st2 in K:
        if (s2, e2, n2) != (s, e, n) and s2 < e + 20000 and e2 > s - 20000:
            tag = "OVERLAP" if (s2 < e and e2 > s) else "near   "
            print(f"    {tag} {n2[:70]:70s} start {s2 - s:+7d} end {e2 - s:+7d} queue {q2} stream {st2}")
