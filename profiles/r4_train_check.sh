#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_train
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_train_step.py tests/test_hip_train.py tests/test_hip_train_step_hybrid.py -x -q -s 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
for v in 0 1; do PRIORFLOW_TRAIN_LOOP=$v timeout -k 10 300 python profiles/time_train_step.py 2>/dev/null | tail -1 | cut -c1-400 | sed "s/^/loop=$v /"; done | tee $O/time.txt
