export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_train_step.py tests/test_hip_kernels.py -x -q -k "sink or unpack or graphed or loop_node or train" 2>&1 | tail -5
python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 | cut -c1-200
PRIORFLOW_GRAD_SINK=0 python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 | cut -c1-200
python profiles/time_train_step.py --steps 10 2>/dev/null | tail -1 | cut -c1-200
