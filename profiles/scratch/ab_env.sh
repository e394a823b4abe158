# same-box A/B of an environment switch:  ab_env.sh VAR=value   (A = unset, B = set), two rounds
export TMPDIR=/tmp
for v in A B A B; do
  mkdir -p gpurun_out/cmp_$v
  if [ $v = B ]; then export "$1"; else unset "${1%%=*}"; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cmp_$v -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "$v: $(python profiles/summarize_trace.py $(find gpurun_out/cmp_$v -name "t_kernel_trace.csv" | head -1) | head -1 | cut -c1-140)"
  rm -rf gpurun_out/cmp_$v
done
