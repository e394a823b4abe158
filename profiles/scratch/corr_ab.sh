export TMPDIR=/tmp
O=gpurun_out/r4_corr
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_fuzz.py -x -q -k "corr or pyramid" 2>&1 | tail -2
for v in 0 1; do
  PRIORFLOW_CORR_XCD2D=$v python3 profiles/ab_corr.py 7 10 tile 2>/dev/null | grep "tile:" | sed "s/^/xcd2d=$v /"
  for c in FETCH_SIZE WRITE_SIZE; do
    PRIORFLOW_CORR_XCD2D=$v rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$v -o t -- python3 profiles/ab_corr.py 1 2 tile > /dev/null 2>&1
  done
  python3 profiles/pmc_traffic.py $O/pmc_FETCH_SIZE_$v $O/pmc_WRITE_SIZE_$v | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['kernels'].items():
    if 'corr' in k: print('xcd2d=$v', k[:60], 'fetch MB', round(v['fetch_bytes_per_launch']/1e6,1), 'write MB', round(v['write_bytes_per_launch']/1e6,1), 'total', round(v['hbm_bytes_per_launch']/1e6,1))"
  rm -rf $O/pmc_FETCH_SIZE_$v $O/pmc_WRITE_SIZE_$v
done
