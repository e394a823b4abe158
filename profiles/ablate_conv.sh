for f in "" prior-flow_amd/lib/diag/*.so; do
  echo "== ${f:-baseline}"
  for w in zr q fh1; do PRIORFLOW_LIB=${f:+$PWD/$f} python profiles/microbench_conv.py 50 $w 2>/dev/null; done
done
