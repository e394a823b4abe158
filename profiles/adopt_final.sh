#!/bin/bash
# Copies the files profiles/refresh_final.sh left under gpurun_out/final6/ into profiles/ as r6_final_<name> (run here, after the
# gpurun call has merged them back) and points profiles/profile_index.json at them.
R=6
for f in gpurun_out/final$R/*; do
  n=$(basename $f)
  case $n in bench_under_rocprof.log|bench_stderr.log|train_trace.log) continue;; esac
  [ -s $f ] && cp $f profiles/r${R}_final_$n
done
python3 - <<PY
import json
p = "profiles/profile_index.json"
d = json.load(open(p))
for e, (st, pm, mf) in zip(d["profiles"], (("kernel_stats.csv", "pmc_traffic.json", "mfma_busy.json"), ("kernel_stats_batch32.csv", None, None))):
    e["round"] = $R
    e["stats"] = "r${R}_final_" + st
    e["pmc"] = ("r${R}_final_" + pm) if pm else None
    e["mfma"] = ("r${R}_final_" + mf) if mf else None
json.dump(d, open(p, "w"), indent=1)
PY
