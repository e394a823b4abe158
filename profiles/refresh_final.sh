#!/bin/bash
# Regenerates the committed round-6 measurement artefacts on the GPU box (run through gpurun from the repo root) into
# gpurun_out/final6/; profiles/adopt_final.sh copies what is to be judged into profiles/ as r6_final_<name> (profiles/profile_index.json names the files
# bench.py reads for `in_replay_us` and `traffic`).  Parts: PART=all | core | train | ab (a gpurun call is limited to 20 minutes).
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/final6
mkdir -p $O
PART=${PART:-all}
B="--no-cpu-baseline --no-batch32 --no-train-step"
if [ $PART = all ] || [ $PART = core ]; then
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
# the bench command under the profiler (bench.py's in_replay_us comes from this summary)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py $B > $O/bench_under_rocprof.log 2>&1
cp $(find $O/prof -name "t_kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 profiles/summarize_trace.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/forward_breakdown.txt
python3 profiles/iteration_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 100 > $O/iteration_timeline.txt
python3 profiles/encoder_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 100 > $O/encoder_timeline.txt
python3 profiles/tail_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 5 > $O/tail_timeline.txt
python3 profiles/chain_gaps.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/chain_gaps.txt     # medians over all steady-state iterations
rm -rf $O/prof
# PMC passes: counters in their own runs, never together with a trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph $B > /dev/null 2>&1
done
python3 profiles/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph $B > /dev/null 2>&1
python3 profiles/mfma_busy.py $O/pmc_mfma > $O/mfma_busy.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
# the bench line quotes `in_replay_us`, `traffic` and the MFMA-busy fractions from the COMMITTED profile files: put this run's
# summaries in their place first (in this box's copy of the tree; profiles/adopt_final.sh does the same at home afterwards), so
# that the committed bench line and the committed profiles are of the same run
R=6
cp $O/kernel_stats.csv profiles/r${R}_final_kernel_stats.csv
cp $O/pmc_traffic.json profiles/r${R}_final_pmc_traffic.json
cp $O/mfma_busy.json profiles/r${R}_final_mfma_busy.json
python bench.py 2> $O/bench_stderr.log | tail -1 > $O/bench_n1.json
python bench.py --batch 32 --steps 5 --warmup 2 --no-train-step 2>/dev/null | tail -1 > $O/bench_batch32.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof32 -o t -- python3 bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-train-step > /dev/null 2>&1
cp $(find $O/prof32 -name "t_kernel_stats.csv" | head -1) $O/kernel_stats_batch32.csv
rm -rf $O/prof32
python profiles/time_sizes.py 512x1024:12 640x1280:32 480x960:12 256x512:4 > $O/time_sizes.txt 2>&1
fi
if [ $PART = all ] || [ $PART = train ]; then
# training step: the captured graph (one graph / split in two), the eager loop node, batch 8, two ranks on this card over gloo
python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 > $O/train_step_time.json
PRIORFLOW_TRAIN_SPLIT_GRAPH=1 python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 > $O/train_step_time_two_graphs.json
python profiles/time_train_step.py --steps 10 2>/dev/null | tail -1 > $O/train_step_time_eager.json
python profiles/time_train_step.py --steps 5 --batch 8 --graph 2>/dev/null | tail -1 > $O/train_step_time_batch8.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 profiles/time_train_step.py --graph --steps 10 2>/dev/null | grep '^{' > $O/train_step_time_2rank_gloo.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/proft -o t -- python3 profiles/time_train_step.py > $O/train_trace.log 2>&1
cp $(find $O/proft -name "t_kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv
python3 profiles/train_sequence.py $(find $O/proft -name "t_kernel_trace.csv" | head -1) > $O/train_sequence.txt     # launches / PyTorch kernels of one eager step
rm -rf $O/proft
rocprofv3 --kernel-trace --output-format csv -d $O/proftg -o t -- python3 profiles/time_train_step.py --steps 3 --warmup 1 --graph > /dev/null 2>&1
python3 profiles/train_phases.py $(find $O/proftg -name "t_kernel_trace.csv" | head -1) > $O/train_phases.txt           # one replayed step by phase
rm -rf $O/proftg
PRIORFLOW_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_gloo2.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tests/run_train_2rank.py 2>&1 | grep -E "rank|checksums" > $O/train_2rank_check.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tests/run_train_2rank.py --graphed 2>&1 | grep -E "rank|graphed" >> $O/train_2rank_check.txt
fi
if [ $PART = all ] || [ $PART = ab ]; then
# same-box A/Bs, interleaved: round 6's split chains (two half-chip chains of one-group launches) against the round-5 schedule,
# and the two corr + pyramid kernels inside the whole forward
profiles/ab_env.sh 3 "PRIORFLOW_SPLIT_AB=0" "PRIORFLOW_SPLIT_AB=1" > $O/ab_split_chains.txt 2>&1
BENCH_ARGS="--batch 32 --no-cpu-baseline --no-batch32 --no-train-step --steps 4 --warmup 2" profiles/ab_env.sh 2 "PRIORFLOW_SPLIT_AB=0" "PRIORFLOW_SPLIT_AB=1" >> $O/ab_split_chains.txt 2>&1
profiles/ab_env.sh 2 "PRIORFLOW_CORR_RS=0" "PRIORFLOW_CORR_RS=1" > $O/ab_corr_form.txt 2>&1
python profiles/ab_corr.py 5 10 tile rs > $O/ab_corr_kernels.txt 2>&1
MB_BATCH=8 python profiles/ab_corr.py 3 4 tile rs >> $O/ab_corr_kernels.txt 2>&1
fi
for f in pytest_gpu.log smoke.log forward_breakdown.txt time_sizes.txt ab_split_chains.txt ab_corr_form.txt ab_corr_kernels.txt train_2rank_check.txt; do [ -f $O/$f ] && { echo "== $f"; tail -12 $O/$f | cut -c1-220; }; done
for f in bench_n1.json bench_batch32.json train_step_time.json train_step_time_two_graphs.json train_step_time_eager.json train_step_time_batch8.json train_step_time_2rank_gloo.json; do [ -f $O/$f ] && { echo "== $f"; cut -c1-330 $O/$f; }; done
exit 0
