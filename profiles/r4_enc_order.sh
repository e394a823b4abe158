#!/bin/bash
# Round 4, step 1: when does cnet start inside the captured graph, and does the capture order move it?
# One rocprofv3 kernel trace per variant; encoder_timeline.py prints the replay's first 2.9 ms.
export TMPDIR=/tmp
O=gpurun_out/r4_enc
mkdir -p $O
./profiles/mfma_rate/mfma_rate > $O/mfma_rate.txt 2>&1
run() {  # name, env assignments...
  name=$1; shift
  ( export "$@"; rocprofv3 --kernel-trace --output-format csv -d $O/prof_$name -o t -- python3 bench.py --no-cpu-baseline --steps 30 ${EXTRA} > $O/log_$name.txt 2>&1 )
  f=$(find $O/prof_$name -name "t_kernel_trace.csv" | head -1)
  python3 profiles/encoder_timeline.py $f 20 3000 > $O/timeline_$name.txt 2>&1
  python3 profiles/summarize_trace.py $f > $O/breakdown_$name.txt 2>&1
  rm -rf $O/prof_$name
  tail -1 $O/log_$name.txt | cut -c1-160
  head -1 $O/breakdown_$name.txt
}
run default PRIORFLOW_DUMMY=1
run cnet_main PRIORFLOW_ENC_ORDER=cnet_main
run both_side PRIORFLOW_ORDER=7
run nofork PRIORFLOW_FORKS=14
EXTRA=--no-graph run eager PRIORFLOW_DUMMY=1
cat $O/mfma_rate.txt
