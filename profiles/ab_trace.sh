export TMPDIR=/tmp
# usage: ab_trace.sh [VARIANT]  -- same-box A/B of the shipped library against prior-flow_amd/lib/diag/lib_VARIANT.so
ALT=${1:-NODMA}
for v in DMA $ALT; do
  lib=$PWD/prior-flow_amd/lib/libpriorflow_hip.so; [ $v = $ALT ] && lib=$PWD/prior-flow_amd/lib/diag/lib_$ALT.so
  mkdir -p gpurun_out/cmp_$v
  PRIORFLOW_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cmp_$v -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python profiles/summarize_trace.py $(find gpurun_out/cmp_$v -name "t_kernel_trace.csv" | head -1) > gpurun_out/cmp_$v.txt
  rm -rf gpurun_out/cmp_$v
done
ALT=$ALT python - <<'PY'
import re, os
ALT=os.environ['ALT']
def load(f):
    d={}
    for l in open(f):
        m=re.match(r'(.*?)\s+n=\s*(\d+)\s+([\d.]+) us', l)
        if m: d[m.group(1)[:75]]=(int(m.group(2)), float(m.group(3)))
    return d, open(f).readline().strip()
a,ha=load('gpurun_out/cmp_DMA.txt'); b,hb=load(f'gpurun_out/cmp_{ALT}.txt')
print('base ', ha[:120]); print(ALT, hb[:120])
for k in sorted(a, key=lambda k:-a[k][1])[:22]:
    if k in b: print(f"{k[28:75]:48s} n={a[k][0]:3d}  base {a[k][1]:8.1f}  {ALT} {b[k][1]:8.1f}  {100*(a[k][1]/b[k][1]-1):+5.1f}%")
PY
