export TMPDIR=/tmp
mkdir -p gpurun_out/r4_chk
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4_chk/pytest.log; cat gpurun_out/r4_chk/pytest.log
for i in 1 2; do for o in fnet_main both_side; do
  PRIORFLOW_ENC_ORDER=$o python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1 enc_order=$o', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done; done | tee gpurun_out/r4_chk/ab_order.txt
python3 bench.py 2> gpurun_out/r4_chk/bench_stderr.log | tail -1 > gpurun_out/r4_chk/bench_n1.json; cut -c1-300 gpurun_out/r4_chk/bench_n1.json
