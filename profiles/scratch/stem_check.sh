export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -k "enc_stem" 2>&1 | tail -5
timeout -k 10 600 python -m pytest tests/test_hip_forward.py tests/test_hip_kernels.py -x -q 2>&1 | tail -4
bash profiles/r4_enc_alone.sh 2>&1 | grep "==\|s2d\|enc_stem\|4, 4" 
for i in 1 2; do for v in 0 1; do PRIORFLOW_STEM_DIRECT=$v python3 bench.py --no-cpu-baseline --no-batch32 --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1 stem_direct=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"; done; done
