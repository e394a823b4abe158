# same-box comparison of PRIORFLOW_ORDER masks: ab_order.sh "1 3 5 9 15"
export TMPDIR=/tmp
for round in 1 2; do
for v in $1; do
  mkdir -p gpurun_out/cmp_$v
  PRIORFLOW_ORDER=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cmp_$v -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "order $v: $(python profiles/summarize_trace.py $(find gpurun_out/cmp_$v -name "t_kernel_trace.csv" | head -1) | head -1 | cut -c1-120)"
  rm -rf gpurun_out/cmp_$v
done; done
