#!/bin/bash
# Training-step evidence of round 4 (profiles/r4_train_step_*): tests, time per step with and without the loop node, kernel trace.
export TMPDIR=/tmp
O=gpurun_out/r4_train_final
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_train_step.py tests/test_hip_train.py tests/test_hip_train_step_hybrid.py -x -q 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
PRIORFLOW_TRAIN_LOOP=1 python profiles/time_train_step.py --steps 10 2>/dev/null | tail -1 > $O/train_step_time.json
PRIORFLOW_TRAIN_LOOP=0 python profiles/time_train_step.py --steps 10 2>/dev/null | tail -1 > $O/train_step_time_tape.json
PRIORFLOW_TRAIN_LOOP=1 python profiles/time_train_step.py --steps 5 --batch 8 2>/dev/null | tail -1 > $O/train_step_time_batch8.json
cut -c1-330 $O/train_step_time.json $O/train_step_time_tape.json $O/train_step_time_batch8.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 profiles/time_train_step.py > $O/log.txt 2>&1
cp $(find $O/prof -name "t_kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv
rm -rf $O/prof
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4_train_final/train_step_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
nat=sum(int(r['Calls']) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'])
natt=sum(float(r['TotalDurationNs']) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'])
print(f"7 steps: {n/7:.0f} launches / step, {tot/7e6:.2f} ms of kernel time / step; at::native + runtime copies: {nat/n:.3f} of the launches, {natt/tot:.3f} of the kernel time")
PY
