#!/usr/bin/env python3
"""Where the chains of a refinement iteration wait: medians over ALL steady-state iterations of a rocprofv3 --kernel-trace CSV of
bench.py (iteration_timeline.py prints one iteration; one iteration of one replay is an anecdote).

For every hardware queue the kernels of an iteration are taken in order (an iteration of a queue starts at its lookup, or at
pf_motion_prep / pf_conf_stem for the two side chains) and each position gets
    dur   the kernel's duration
    gap   its start minus the end of the previous kernel ON THE SAME QUEUE (idle time of that chain)
    dep   its start minus the latest end, in the same iteration, of any kernel on ANOTHER queue that ended before it started
          (how long ago the last possible cross-queue producer had finished: small = the chain was waiting for that queue)
The first and the last iteration of every forward are left out (corr build in front, mask head behind).
usage: chain_gaps.py <kernel_trace.csv>"""
import csv
import statistics
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if "PfLookupArgs" in n:
        return "pf_lookup"
    return n.split("(")[0][:34]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), str(r.get("Queue_Id", "?")), short(r["Kernel_Name"])) for r in rows]
    ks.sort()
    # forwards: split at the first corr launch of each replay
    fw, cur = [], None
    for k in ks:
        if "pf_corr_rs" in k[3] or "pf_corr_mfma" in k[3]:
            if cur is not None and cur and not ("pf_corr" in cur[-1][3]):
                fw.append(cur)
                cur = []
            if cur is None:
                cur = []
        if cur is not None:
            cur.append(k)
    fw = fw[2:]                                              # warm-up replays
    per = {}                                                 # (chain, pos) -> lists
    spans = []
    for f in fw:
        mp = [k for k in f if k[3].startswith("pf_motion_prep")]
        if len(mp) < 4:
            continue
        spans += [(b[0] - a[0]) / 1e3 for a, b in zip(mp[1:-2], mp[2:-1])]
        queues = sorted({k[2] for k in f})
        for q in queues:
            mine = [k for k in f if k[2] == q]
            heads = [i for i, k in enumerate(mine) if k[3] in ("pf_lookup", "pf_motion_prep_kernel", "pf_conf_stem_kernel")]
            # an iteration of this queue: from one head kernel of the FIRST kind seen to the next of that kind
            if not heads:
                continue
            kind = mine[heads[0]][3]
            starts = [i for i in heads if mine[i][3] == kind]
            for a, b in zip(starts[1:-2], starts[2:-1]):
                seq = mine[a:b]
                names = tuple(k[3] for k in seq)
                chain = ("B" if any("<2, 3, 3, 2>" in n for n in names) else "A") if kind == "pf_lookup" else \
                        ("flow" if kind.startswith("pf_motion") else "conf")
                t_lo, t_hi = mine[a - 1][1] if a else seq[0][0], seq[-1][1]
                others = [k for k in f if k[2] != q and t_lo - 400_000 < k[1] < t_hi]
                for pos, k in enumerate(seq):
                    prev_end = mine[a + pos - 1][1] if a + pos else k[0]
                    cands = [o[1] for o in others if o[1] <= k[0]]
                    dep = (k[0] - max(cands)) / 1e3 if cands else float("nan")
                    d = per.setdefault((chain, pos, k[3]), [[], [], []])
                    d[0].append((k[1] - k[0]) / 1e3)
                    d[1].append((k[0] - prev_end) / 1e3)
                    d[2].append(dep)
    print(f"# {len(fw)} forwards, {len(spans)} steady-state iterations; span median {statistics.median(spans):.1f} us "
          f"(p10 {sorted(spans)[len(spans) // 10]:.1f}, p90 {sorted(spans)[9 * len(spans) // 10]:.1f})")
    for chain in ("A", "B", "flow", "conf"):
        keys = sorted(k for k in per if k[0] == chain)
        if not keys:
            continue
        tot_d = tot_g = 0.0
        print(f"chain {chain}:   pos  kernel                              n     dur     gap   dep(last other-queue end before start)")
        for k in keys:
            d = per[k]
            if len(d[0]) < max(3, len(spans) // 4):
                continue
            md, mg = statistics.median(d[0]), statistics.median(d[1])
            deps = [x for x in d[2] if x == x]
            tot_d += md
            tot_g += mg
            print(f"          {k[1]:5d}  {k[2]:34s} {len(d[0]):5d} {md:7.1f} {mg:7.1f} {statistics.median(deps) if deps else float('nan'):7.1f}")
        print(f"          sum of durations {tot_d:.1f} us, of gaps {tot_g:.1f} us")


if __name__ == "__main__":
    main()
