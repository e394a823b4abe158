#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_enc_alone
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 profiles/enc_alone.py > $O/log.txt 2>&1
python3 profiles/enc_alone_report.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/enc_alone.txt 2>&1
rm -rf $O/prof
cat $O/enc_alone.txt
