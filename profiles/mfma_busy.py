#!/usr/bin/env python3
"""Matrix-pipe busy fraction per kernel from a rocprofv3 PMC pass:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
  python profiles/mfma_busy.py <dir>
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16, summed over the SIMDs); GRBM_GUI_ACTIVE is summed over
the 8 XCDs (MI355X_MICROARCH.md).  busy fraction = MFMA_BUSY / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import json
import sys

import statistics

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    if "pf_conv" not in k and "pf_corr" not in k and "pf_combine_conv" not in k and "pf_enc_stem" not in k and "pf_enc_conv" not in k:
        continue
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    # MEDIAN over the launches (round 3 used the mean: the first corr-build launch of a process counted 4x the GRBM cycles of the
    # others -- first-touch page faults of the 373 MB volume -- and the kernel's entry read 0.071 instead of ~0.27)
    n = len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    busy = statistics.median(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = statistics.median(c["GRBM_GUI_ACTIVE"])
    out[k[:110]] = {"launches": n, "mfma_busy_cycles_per_launch": busy, "grbm_gui_active_per_launch": gui,
                    "mfma_busy_fraction": busy / (1024.0 * gui / 8.0)}
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, bench.py --no-graph (single stream), B=1 512x1024 iters=12",
           "kernels": out}, sys.stdout, indent=1)
