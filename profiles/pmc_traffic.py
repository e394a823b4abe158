#!/usr/bin/env python3
"""HBM-side traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: MI355X_MICROARCH.md 'rocprofv3 PMC slots'):

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rd -o t -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_wr -o t -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
  python profiles/pmc_traffic.py gpurun_out/pmc_rd gpurun_out/pmc_wr > profiles/r2_pmc_traffic.json

Units / corrections (same guide, section HBM): both counters are KiB derived from the L2's memory-side
request counters (Infinity-Cache hits are counted, not excluded); on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads, so it is doubled.  Output: per kernel name the mean bytes per launch."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


rd, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(rd) | set(wr)):
    # every kernel of the library (round 5 filtered by a list of substrings; a kernel that shipped after the list was written --
    # pf_enc_conv64_kernel -- then had no entry and the bench line's `traffic` was null: VERDICT r5)
    if "pf_" not in k and "Pf" not in k:
        continue
    n = max(rd.get(k, [0])[0], wr.get(k, [0])[0])
    fetch = 2.0 * 1024.0 * rd[k][1] / rd[k][0] if k in rd and rd[k][0] else None
    write = 1024.0 * wr[k][1] / wr[k][0] if k in wr and wr[k][0] else None
    out[k[:110]] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "hbm_bytes_per_launch": (fetch or 0.0) + (write or 0.0)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --no-graph, B=1 512x1024 iters=12; "
                     "FETCH_SIZE x2 (gfx950), KiB -> bytes", "kernels": out}, sys.stdout, indent=1)
