#!/bin/bash
# usage: trace_variants.sh "<lib variants under prior-flow_amd/lib/diag (without lib_/.so)>" "<microbench cases>"
export TMPDIR=/tmp
for v in $1; do for w in $2; do
  lib=$PWD/prior-flow_amd/lib/diag/lib_$v.so; [ "$v" = base ] && lib=$PWD/prior-flow_amd/lib/libpriorflow_hip.so
  PRIORFLOW_LIB=$lib rocprofv3 --kernel-trace -d gpurun_out/tv_${v}_$w -o t -- python profiles/microbench_conv.py 50 $w > /dev/null 2>&1
  python - "$v" "$w" <<PY
import sqlite3, glob, sys
v, w = sys.argv[1:3]
for f in glob.glob(f"gpurun_out/tv_{v}_{w}/**/t_results.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
    for r in db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name"):
        if 'conv' in r[0]: print(f"{v:24s} {w:4s} calls {r[1]:3d} avg {r[2]/1e3:7.1f} us  min {r[3]/1e3:7.1f} us")
PY
  rm -rf gpurun_out/tv_${v}_$w
done; done
